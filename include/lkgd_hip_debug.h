/* lkgd_hip_debug.h - debug / measurement knobs of liblkgd_hip.so.  NOT part of the reference-facing interface
 * (include/lkgd_hip.h); included by the A/B tools under tools/ and by the per-variant parity tests only.
 *
 * Every setter writes a THREAD-LOCAL variable of the library (round 6; process-global until round 5): the knob applies to the
 * launches the calling host thread makes afterwards and to nobody else's - a second pipeline on another thread keeps the
 * automatic behaviour (tests/test_host_cpu.py::test_debug_knobs_are_per_thread).  Value 0 (or -1 where noted) restores the
 * automatic behaviour.
 *      lkgd_debug_set_gemm_variant(v)   force a tile program of lkgd_gemm_f16 where it applies: 1 = 128x128, 2 = 256x128 ring,
 *                                       7 = 128x128 on a four-stage ring (few-row problems), 3 = persistent 256x128, 4 = 256x320,
 *                                       5 = row-panel, 6 = resident-weight; 0 = auto
 *      lkgd_debug_set_gemm_splitk(on)   0 = never cut K into slices
 *      lkgd_debug_set_mid_model(tk, fix, red, tbs, forced)   cost model of the four-stage 128x128 program's K slicing (us per
 *                                       K-tile, us per workgroup, us per reduce pass, TB/s of the reduce pass; <= 0 keeps a value);
 *                                       forced > 0 = that many slices where legal
 *      lkgd_debug_set_wide_ksplit(k)    force k K-slices on the 256x320 program where legal; 0 = rule
 *      lkgd_debug_set_wide_lds_out(on)  256x320 program: 0 = direct 8-byte stores everywhere, 1 / -1 = rows through LDS where used
 *      lkgd_debug_set_wide_tile_n(wn)   256x320 program: force 256 / 320 tile columns where that width divides N; 0 = the rule
 *      lkgd_debug_set_wide_tile_m(wm)   256x320 program: force 192 / 256 tile rows on unsliced launches; 0 = the rule
 *      lkgd_debug_set_attn_waves(nw)    spatial attention: waves per workgroup (4 / 8 / 16); 0 = by sequence length
 *      lkgd_debug_set_attn_kvb(kvb)     spatial attention: keys per barrier (64 / 128); 0 = default
 *      lkgd_debug_set_attn_pipe(mode)   spatial attention: 1 = never the software-pipelined program (attn_spatial_pipe.hip),
 *                                       2 = wherever it is legal (S >= 128; S % 128 != 0 runs its masked form); 0 = by sequence length
 *      lkgd_debug_set_gn_apply_kb(kb) / lkgd_debug_set_gn_stats_kb(kb)   GroupNorm chunk sizes in KiB (>= 32)
 *      lkgd_debug_set_gn_target_wgs(n)  workgroups a GroupNorm pass aims at on small maps (chunks shrink to 8 KiB); 1 = fixed sizes
 *      lkgd_debug_set_gn_small(on)      0 = lkgd_groupnorm_silu always takes the three launches
 *      lkgd_debug_set_gn_small_limits(bytes)   tensor size up to which the one-launch form is taken
 */
#ifndef LKGD_HIP_DEBUG_H
#define LKGD_HIP_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

void lkgd_debug_set_gemm_variant(int32_t v);
void lkgd_debug_set_gemm_splitk(int32_t on);
void lkgd_debug_set_mid_model(float tk_us, float fix_us, float red_us, float red_tbs, int32_t forced);
void lkgd_debug_set_wide_ksplit(int32_t k);
void lkgd_debug_set_wide_lds_out(int32_t on);
void lkgd_debug_set_wide_tile_n(int32_t wn);
void lkgd_debug_set_wide_tile_m(int32_t wm);
void lkgd_debug_set_attn_waves(int32_t nw);
void lkgd_debug_set_attn_kvb(int32_t kvb);
void lkgd_debug_set_attn_pipe(int32_t mode);
void lkgd_debug_set_gn_apply_kb(int32_t kb);
void lkgd_debug_set_gn_stats_kb(int32_t kb);
void lkgd_debug_set_gn_target_wgs(int32_t n);
void lkgd_debug_set_gn_small(int32_t on);
void lkgd_debug_set_gn_small_limits(int64_t total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LKGD_HIP_DEBUG_H */
