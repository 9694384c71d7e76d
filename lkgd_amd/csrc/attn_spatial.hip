// Spatial self-attention, head_dim 64, fp16 in / fp32 softmax+accumulate (include/lkgd_hip.h section 4).
//
// Flash-style for gfx950: one workgroup = NW waves x 32 query rows (NW = 4, 8 or 16 by sequence length: a staged K/V tile
// serves 32*NW queries); K/V tiles of 64 keys stream through an
// LDS ring filled by LDS-DMA (global_load_lds_dwordx4, the XOR swizzles applied on the source side, no staging
// registers: 120 VGPRs, four waves per SIMD): the next tile is issued right after the barrier that opens a tile; one
// barrier per tile.  Measured on this kernel (tools/micro/attn_pmc.sh, attn_lib.py): VALU issue 62 % and MFMA 35 % of
// SIMD time, a wave is stalled two thirds of its life; removing every exp changes the time by 6 %, a deeper K/V ring by
// nothing, interleaving the MFMA accumulators by nothing: the per-wave chain QK^T -> max -> exp -> PV is latency-bound
// and only more resident waves help.
//   S^T = K . Q^T   "swapped" product: v_mfma_f32_32x32x16_f16 with A = K rows (ds_read_b128 from an XOR-swizzled
//                   [key][64] image) and B = Q^T held in registers, so a lane owns ONE query column and 32 of the
//                   tile's 64 scores: row max / row sum are 31 register ops + one cross-half shuffle.
//   O^T += V^T . P^T  the S^T accumulator, converted to fp16 in place, IS the B operand (k = key on the lane-half /
//                   register index); A = V^T comes from the row-major V image through ds_read_b64_tr_b16
//                   (hardware-transposed LDS read), laid out so the 4-row x 16-col blocks are bank-conflict free.
// Softmax scale and log2(e) are folded into one FMA feeding v_exp_f32 (exp2).
#include "common.h"

#define KVBLK 64
#ifndef ATT_NST
#define ATT_NST 2   // 2 stages (32 KiB) keep four workgroups per CU; a third stage measured 4 % slower (occupancy 3)
#endif
#define ATT_LDS (ATT_NST * 2 * KVBLK * 64 * 2)  // 3 stages x (K 8 KiB + V 8 KiB)

__device__ __forceinline__ int k_lds_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_lds_off(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4); }

// NW = waves per workgroup (32 query rows each): a K/V tile staged once serves 32*NW queries, so NW = 8 halves the
// L2 -> LDS traffic per flop of NW = 4 (11 instead of 22 bytes per clock and CU at S = 9216)
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 1 : 2) void attn_spatial_kernel(const half_t* __restrict__ q, int ldq,
                                                              const half_t* __restrict__ k, int ldk,
                                                              const half_t* __restrict__ v, int ldv,
                                                              half_t* __restrict__ out, int ldo, int S, int heads,
                                                              const int* __restrict__ kvmap, float scale_log2e,
                                                              int nqb, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  // XCD-aware bijective remap: blocks of one (batch, head) stay on one XCD so K/V are served from its L2
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, slot = bid >> 3;
    int qq = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
  }
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int n = bh / heads, head = bh - n * heads;
  const int kvn = kvmap ? kvmap[n] : n;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[qrow][16*ks + 8*h + 0..7]
  constexpr int QBLK = NW * 32;
  const int qrow = qb * QBLK + w * 32 + l31;
  const int qrow_c = qrow < S ? qrow : S - 1;
  const half_t* qp = q + ((long long)n * S + qrow_c) * ldq + head * 64 + h * 8;
  half8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8_t*)(qp + ks * 16);

  // ---- staging map: thread t fills 16-byte chunk c = t & 7 of LDS rows srow0 and srow0 + 32, for K and for V.  LDS-DMA
  //      writes lane-linearly (wave base + lane*16 = 8 rows x 8 chunks per wave), so the chunk a position holds is
  //      chosen on the SOURCE side: position c of row r holds source chunk c ^ swizzle(r)
  const half_t* kbase = k + (long long)kvn * S * ldk + head * 64;
  const half_t* vbase = v + (long long)kvn * S * ldv + head * 64;
  const int srow0 = t >> 3, sc = t & 7;
  // (NW = 8: 512 threads cover the 64 rows of a tile in one pass; NW = 4: rows srow0 and srow0 + 32)
  const int kc0 = (sc ^ ((srow0 >> 1) & 7)) * 8, kc1 = (sc ^ (((srow0 + 32) >> 1) & 7)) * 8;
  const int vc0 = (sc ^ (((srow0 >> 1) & 1) << 2)) * 8, vc1 = (sc ^ ((((srow0 + 32) >> 1) & 1) << 2)) * 8;
#define ISSUE_TILE(tile, st)                                                   \
  {                                                                            \
    int key0_ = (tile) * KVBLK + srow0, key1_ = key0_ + 32;                    \
    if (key0_ >= S) key0_ = S - 1; /* clamped rows are masked in the scores */ \
    if (key1_ >= S) key1_ = S - 1;                                             \
    char* kb_ = smem + (st) * (2 * KVBLK * 128) + w * 1024;                    \
    if (NW <= 8 || w < 8) glds16(kbase + (long long)key0_ * ldk + kc0, kb_);   \
    if (NW == 4) glds16(kbase + (long long)key1_ * ldk + kc1, kb_ + 32 * 128); \
    if (NW <= 8 || w < 8) glds16(vbase + (long long)key0_ * ldv + vc0, kb_ + KVBLK * 128); \
    if (NW == 4) glds16(vbase + (long long)key1_ * ldv + vc1, kb_ + KVBLK * 128 + 32 * 128); \
  }

  float16_t oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (S + KVBLK - 1) / KVBLK;
  ISSUE_TILE(0, 0);
  if (ATT_NST > 2 && ntiles > 1) ISSUE_TILE(1, 1);

  // transposed-read lane constants: 16-lane group -> (h, dgrp); lane in group i -> (row q4 = i>>2, col part = i&3)
  const int i16 = lane & 15;
  const int dgrp = (lane >> 4) & 1;
  const int tr_row = 4 * h + (i16 >> 2);
  const int tr_c = dgrp * 2 + ((i16 & 3) >> 1);   // 16-byte chunk inside the 32-column d-fragment
  const int tr_sub = (i16 & 1) * 8;

  int cur = 0;
  for (int j = 0; j < ntiles; ++j) {
    // tile j has landed (this thread's four loads; the barrier publishes everyone's) - tile j+1 may stay in flight - and
    // every wave is done with tile j-1, whose stage takes tile j+2
    if (ATT_NST > 2 && j + 1 < ntiles) { if (NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + ATT_NST - 1 < ntiles) {
      const int nst = cur == 0 ? ATT_NST - 1 : cur - 1;  // (j + ATT_NST - 1) % ATT_NST
      ISSUE_TILE(j + ATT_NST - 1, nst);
    }
    const char* kb = smem + cur * (2 * KVBLK * 128);
    const char* vb = kb + KVBLK * 128;

    // ---- S^T tile: 64 keys x 32 queries per wave
    // (K-step outer, key fragment inner: consecutive MFMAs alternate between the two accumulators instead of forming
    // two chains of four back-to-back dependent ones)
    float16_t s[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        half8_t kf = *(const half8_t*)(kb + k_lds_off(32 * f + l31, ks * 2 + h));
        s[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[f], 0, 0, 0);
      }
    }
    if ((j + 1) * KVBLK > S) {   // ragged last tile: mask keys >= S
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = j * KVBLK + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (key >= S) s[f][r] = -1e30f;
        }
    }
    // ---- online softmax (per query column; the two lane halves hold disjoint keys of the same query)
    // row max as a tree of 3-input maxima (v_max3_f32: 16 instead of 32 issue slots)
    float mx0 = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
    float mx1 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
      mx0 = fmaxf(fmaxf(mx0, s[0][r]), s[0][r + 1]);
      mx1 = fmaxf(fmaxf(mx1, s[1][r]), s[1][r + 1]);
    }
    float mx = fmaxf(fmaxf(mx0, mx1), fmaxf(s[0][15], s[1][15]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const bool grew = m_new > m_run;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);   // == 1 where the max did not grow
    const float mb = m_new * scale_log2e;
    m_run = m_new;
    float psum = 0.f;
    half8_t pf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        half8_t pv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[f][ss * 8 + e], scale_log2e, -mb));
          psum += p;
          pv[e] = (half_t)p;
        }
        pf[f][ss] = pv;
      }
    l_run = l_run * alpha + psum;
    // rescale O only when some query of this wave raised its running max (rare after the first tiles): wave-uniform
    if (__any(grew)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }

    // ---- O^T += V^T . P^T : k-slot (half h, element jj) of k-step (f, ss) is key 32f + 16ss + 8(jj>>2) + 4h + (jj&3)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int df = 0; df < 2; ++df) {     // d-fragment innermost: the two output accumulators alternate
          int row0 = 32 * f + 16 * ss + tr_row;
          fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(vb + v_lds_off(row0, df * 4 + tr_c) + tr_sub));
          fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4_t*)(vb + v_lds_off(row0 + 8, df * 4 + tr_c) + tr_sub));
          half4_t lo4 = __builtin_bit_cast(half4_t, lo), hi4 = __builtin_bit_cast(half4_t, hi);
          half8_t vf = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[f][ss], oacc[df], 0, 0, 0);
        }
    }
    cur = cur == ATT_NST - 1 ? 0 : cur + 1;
  }

  // ---- normalise and store: lane owns query row qrow, d = 32*df + 8*(r>>2) + 4*h + (r&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < S) {
    half_t* op = out + ((long long)n * S + qrow) * ldo + head * 64 + 4 * h;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(oacc[df][g * 4 + e] * inv);
        *(half4_t*)(op + 32 * df + 8 * g) = o;
      }
  }
}

static int attn_nw_override = 0;     // A/B knob: 4 or 8 waves per workgroup regardless of S
extern "C" void lkgd_debug_set_attn_waves(int nw) { attn_nw_override = nw; }

extern "C" int lkgd_attn_spatial(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                 void* out, int32_t ldo, int32_t nbatch, int32_t S, int32_t heads,
                                 const int32_t* kv_batch_map, float scale, lkgd_stream_t stream) {
  if (!q || !k || !v || !out) return LKGD_E_NULL;
  if (nbatch <= 0 || S <= 0 || heads <= 0) return LKGD_E_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return LKGD_E_ALIGN;
  if (ldq < heads * 64 || ldk < heads * 64 || ldv < heads * 64 || ldo < heads * 64) return LKGD_E_SHAPE;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || ((uintptr_t)out & 7)) return LKGD_E_ALIGN;
  // queries per workgroup: 512 at S >= 8192, 256 at S >= 2304, else 128 (tools/attn_bench.py with ATTN_WAVES: S = 9216
  // 4.03 / 3.89 / 3.78 ms for 128 / 256 / 512; S = 2304 0.507 / 0.495 / 0.55; S = 576 0.083 / 0.100: partial last blocks)
  const int nw = attn_nw_override ? attn_nw_override : (S >= 8192 ? 16 : S >= 2304 ? 8 : 4);
  const int QBLK = nw * 32;
  int nqb = (S + QBLK - 1) / QBLK;
  long long nwg = (long long)nqb * nbatch * heads;
  if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
  if (nw == 16)
    hipLaunchKernelGGL(attn_spatial_kernel<16>, dim3((unsigned)nwg), dim3(1024), ATT_LDS, (hipStream_t)stream,
                       (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, S, heads,
                       kv_batch_map, scale * 1.4426950408889634f, nqb, (int)nwg);
  else if (nw == 8)
    hipLaunchKernelGGL(attn_spatial_kernel<8>, dim3((unsigned)nwg), dim3(512), ATT_LDS, (hipStream_t)stream,
                       (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, S, heads,
                       kv_batch_map, scale * 1.4426950408889634f, nqb, (int)nwg);
  else
    hipLaunchKernelGGL(attn_spatial_kernel<4>, dim3((unsigned)nwg), dim3(256), ATT_LDS, (hipStream_t)stream,
                       (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, S, heads,
                       kv_batch_map, scale * 1.4426950408889634f, nqb, (int)nwg);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
