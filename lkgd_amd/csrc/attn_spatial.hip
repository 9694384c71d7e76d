// Spatial self-attention, head_dim 64, fp16 in / fp32 softmax+accumulate (include/lkgd_hip.h section 4).
//
// Flash-style for gfx950: one workgroup = NW waves x 32 query rows (NW = 4, 8 or 16 by sequence length: a staged K/V tile
// serves 32*NW queries); K/V tiles stream through a two-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4, the XOR
// swizzles applied on the source side, no staging registers: <= 127 VGPRs, four waves per SIMD): the next tile is issued
// right after the barrier that opens a tile; one barrier per tile.
//   S^T = K . Q^T   "swapped" product: v_mfma_f32_32x32x16_f16 with A = K rows (ds_read_b128 from an XOR-swizzled
//                   [key][64] image) and B = Q^T held in registers, so a lane owns ONE query column and 32 of the
//                   tile's 64 scores: row max / row sum are register ops, a cross-half shuffle only when the reference moves.
//   O^T += V^T . P^T  the S^T accumulator, converted to fp16 in place, IS the B operand (k = key on the lane-half /
//                   register index); A = V^T comes from the row-major V image through ds_read_b64_tr_b16
//                   (hardware-transposed LDS read), laid out so the 4-row x 16-col blocks are bank-conflict free
//                   (SQ_LDS_BANK_CONFLICT = 0, profiles/r02_pmc_attn_spatial.txt).
// What bounds it (head_dim 64): per 64-key tile a wave issues 18 MFMAs (576 matrix-pipe cycles) and ~600 cycles of VALU, 264 of
// them the 33 v_exp_f32 (8 issue cycles each) - two exponentials per MFMA gap where the matrix pipe leaves room for one, so
// the kernel is VALU-issue-bound, not matrix-bound; LDS is 26 % active.  Round 2 took the multiply-subtract in front of
// every exponential and the per-tile rescale out of the VALU stream (below): 3.78 -> 3.50 ms at S = 9216, 0.50 -> 0.42 at 2304.
#include "common.h"

#define KVBLK 64            // keys per MFMA sub-tile
#ifndef ATT_KVB16
#define ATT_KVB16 128        // keys per barrier of the 16-wave kernel
#endif
#ifndef ATT_NST
#define ATT_NST 2           // LDS stages of KVB keys each (K image + V image per stage); 3 = the loads get two tiles of flight time
#endif
#define ATT_THR 5.0f        // the running reference max moves only when a score exceeds it by more than this (log2 units)

__device__ __forceinline__ int k_lds_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_lds_off(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4); }

// Also measured and dropped in round 2: TWO 32-query tiles per wave (eight waves, two per SIMD, 232-256 registers; every K / V^T
// fragment read from LDS feeding two MFMAs, two independent softmax chains per wave): bit-identical results, 893 instead of
// 965 TFLOP/s at S = 9216 and 758 instead of 910 at S = 2304 - halving the resident waves costs more than the halved LDS reads
// and the in-wave overlap return.
// NW  = waves per workgroup (32 query rows each): a K/V tile staged once serves 32*NW queries
// KVB = keys staged per barrier (64 or 128): two 64-key sub-tiles per barrier halve the lockstep points of the 16 waves
//
// Softmax bookkeeping (VALU is the co-bound pipe at head_dim 64: 32 exponentials per lane and tile against 16 MFMAs):
//   * Q is pre-multiplied by scale*log2(e) when it is loaded, so the accumulators are in exp2 units;
//   * the reference maximum mb of a query is SUBTRACTED BY THE MATRIX PIPE: a fifth k-step with A = e_0 (1.0 in k-slot 0)
//     and B = (-mb in k-slot 0) runs first with C = 0, so the scores arrive as s - mb and exp2 needs no FMA in front;
//     mb is kept exactly fp16-representable, the product 1.0 * (-mb) is exact;
//   * mb moves only when some score of the wave exceeds it by more than ATT_THR (wave-uniform, rare after the first
//     tiles): then the scores, O and l are brought to the new reference exactly once, before any exponential of the tile
//     (probabilities stay <= 2^ATT_THR, well inside fp16, and relative precision does not depend on the reference).
template <int NW, int KVB>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 1 : 2) void attn_spatial_kernel(const half_t* __restrict__ q, int ldq,
                                                              const half_t* __restrict__ k, int ldk,
                                                              const half_t* __restrict__ v, int ldv,
                                                              half_t* __restrict__ out, int ldo, int Sq, int S, int heads,
                                                              const int* __restrict__ kvmap, float scale_log2e,
                                                              int nqb, int nwg) {
  // Sq query rows per batch entry (q / out), S key rows per batch entry (k / v): equal except for a frame-sharded DiT rank,
  // whose local queries attend to the gathered keys of all ranks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  constexpr int STAGE = 2 * KVB * 128;          // bytes: K image then V image
  constexpr int NSUB = KVB / KVBLK;
  constexpr int ROWS_PER_PASS = NW * 8;         // a wave's LDS-DMA instruction covers 8 rows x 8 chunks
  constexpr int NPASS = (KVB + ROWS_PER_PASS - 1) / ROWS_PER_PASS;

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

  // ---- Q fragments (B operand of S^T = K.Q^T), pre-scaled: lane holds Q[qrow][16*ks + 8*h + 0..7] * scale*log2e
  constexpr int QBLK = NW * 32;
  const int qrow = qb * QBLK + w * 32 + l31;
  const int qrow_c = qrow < Sq ? qrow : Sq - 1;
  const half_t* qp = q + ((long long)n * Sq + qrow_c) * ldq + head * 64 + h * 8;
  half8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const half8_t raw = *(const half8_t*)(qp + ks * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = (half_t)((float)raw[e] * scale_log2e);
  }

  // ---- staging map: thread t fills 16-byte chunk sc = t & 7 of tile rows p*ROWS_PER_PASS + (t >> 3), for K and for V.
  //      LDS-DMA writes lane-linearly (wave base + lane*16 = 8 rows x 8 chunks per wave), so the chunk a position holds is
  //      chosen on the SOURCE side: position c of row r holds source chunk c ^ swizzle(r)  (r mod 16 is pass-invariant)
  const half_t* kbase = k + (long long)kvn * S * ldk + head * 64;
  const half_t* vbase = v + (long long)kvn * S * ldv + head * 64;
  const int srow0 = t >> 3, sc = t & 7;
  const int kc0 = (sc ^ ((srow0 >> 1) & 7)) * 8;
  const int vc0 = (sc ^ (((srow0 >> 1) & 1) << 2)) * 8;
#define ISSUE_TILE(tile, st)                                                              \
  {                                                                                       \
    _Pragma("unroll") for (int p_ = 0; p_ < NPASS; ++p_) {                                \
      if (p_ * ROWS_PER_PASS + w * 8 < KVB) { /* wave-uniform */                          \
        int key_ = (tile) * KVB + p_ * ROWS_PER_PASS + srow0;                             \
        if (key_ >= S) key_ = S - 1; /* clamped rows are masked in the scores */          \
        char* kb_ = smem + (st) * STAGE + (p_ * ROWS_PER_PASS + w * 8) * 128;             \
        glds16(kbase + (long long)key_ * ldk + kc0, kb_);                                 \
        glds16(vbase + (long long)key_ * ldv + vc0, kb_ + KVB * 128);                     \
      }                                                                                   \
    }                                                                                     \
  }

  float16_t oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float mb = 0.f, l_run = 0.f;      // reference max (exp2 units, fp16-representable) and running sum of this lane half
  // bias k-step operands: A = 1.0 in k-slot 0 of every key row, B = -mb in k-slot 0 of the lane's query column
  half8_t abias, bbias;
#pragma unroll
  for (int e = 0; e < 8; ++e) { abias[e] = (half_t)0.f; bbias[e] = (half_t)0.f; }
  if (h == 0) abias[0] = (half_t)1.f;

  const int ntiles = (S + KVB - 1) / KVB;
  ISSUE_TILE(0, 0);
#if ATT_NST == 3
  if (ntiles > 1) ISSUE_TILE(1, 1);
#endif

  // transposed-read lane constants: 16-lane group -> (h, dgrp); lane in group i -> (row q4 = i>>2, col part = i&3)
  const int i16 = lane & 15;
  const int dgrp = (lane >> 4) & 1;
  const int tr_row = 4 * h + (i16 >> 2);
  const int tr_c = dgrp * 2 + ((i16 & 3) >> 1);   // 16-byte chunk inside the 32-column d-fragment
  const int tr_sub = (i16 & 1) * 8;

  int cur = 0;
  for (int j = 0; j < ntiles; ++j) {
    // tile j has landed (this thread's loads; the barrier publishes everyone's) and every wave is done with tile j-1,
    // whose stage takes tile j+1
#ifndef ATT_X_NOSYNC      // timing knob (tools/micro/attn_knobs.sh): no DMA stream, no barrier - every tile reads stage 0
#if ATT_NST == 3
    // tile j's loads are older than tile j+1's (2 per pass this wave takes part in): those may stay in flight
    if (j + 1 < ntiles) {
      if (NPASS == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + 2 < ntiles) ISSUE_TILE(j + 2, (cur + 2) % 3);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + 1 < ntiles) ISSUE_TILE(j + 1, cur ^ 1);
#endif
#endif
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
      if (NSUB > 1 && (j * KVB + sub * KVBLK) >= S) break;      // wave-uniform: the ragged tail has no second sub-tile
      const char* kb = smem + cur * STAGE + sub * (KVBLK * 128);
      const char* vb = kb + KVB * 128;
      const int key_base = j * KVB + sub * KVBLK;

      // ---- S^T - mb: 64 keys x 32 queries per wave; the bias k-step first (C = 0), then K-step outer / key fragment inner
      float16_t s[2];
      {
        float16_t z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abias, bbias, z, 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abias, bbias, z, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#ifdef ATT_X_NOKREAD     // timing knob: no K fragment reads
          half8_t kf = qf[(ks + f) & 3];
#else
          half8_t kf = *(const half8_t*)(kb + k_lds_off(32 * f + l31, ks * 2 + h));
#endif
#ifndef ATT_X_NOQK
          s[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[f], 0, 0, 0);
#else
          s[f][ks] += (float)kf[0];
#endif
        }
      }
      if (key_base + KVBLK > S) {   // ragged last sub-tile: mask keys >= S
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int key = key_base + 32 * f + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key >= S) s[f][r] = -1e30f;
          }
      }
      // ---- does the reference have to move?  per-lane max as a tree of 3-input maxima, one wave vote
#ifdef ATT_X_NOMAX
      float mx = s[0][0];
#else
      float mx0 = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
      float mx1 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) {
        mx0 = fmaxf(fmaxf(mx0, s[0][r]), s[0][r + 1]);
        mx1 = fmaxf(fmaxf(mx1, s[1][r]), s[1][r + 1]);
      }
      float mx = fmaxf(fmaxf(mx0, mx1), fmaxf(s[0][15], s[1][15]));
#endif
      const bool first = (j == 0 && sub == 0);
      // (Tried and dropped in round 2: not computing the tile maximum at all and voting on the row sum instead, redoing the
      // tile from LDS in the rare case - the exponentials then all have to finish before the vote, the compiler can no longer
      // interleave them with the P.V MFMAs, and the kernel needs 152 registers: 3 instead of 4 waves per SIMD.)
#ifdef ATT_X_NOMAX       // timing knob: no per-tile maximum / vote
      if (first) {
#else
      if (first || __any(mx > ATT_THR)) {
#endif
        // (the two lane halves hold disjoint keys of the same query: one reference per query)
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float tgt = mb + mx;                                  // the tile's true maximum
        if (!first) tgt = fmaxf(tgt, mb);                     // the reference never moves down after the first tile
        tgt = fminf(fmaxf(tgt, -60000.f), 60000.f);
        const float mb_new = (float)(half_t)tgt;              // keep it exactly representable for the bias k-step
        const float delta = mb_new - mb;
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);      // O and l are zero at the first tile
        mb = mb_new;
        bbias[0] = h == 0 ? (half_t)(-mb_new) : (half_t)0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[f][r] -= delta;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      }
      // ---- probabilities: exp2 straight off the accumulators
      float psum = 0.f;
      half8_t pf[2][2];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          half8_t pv;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
#ifdef ATT_X_NOEXP       // timing knob: no exponentials
            float p = s[f][ss * 8 + e];
#else
            float p = __builtin_amdgcn_exp2f(s[f][ss * 8 + e]);
#endif
#ifndef ATT_X_NOSUM
            psum += p;
#endif
            pv[e] = (half_t)p;
          }
          pf[f][ss] = pv;
        }
      l_run += psum;

      // ---- O^T += V^T . P^T : k-slot (half h, element jj) of k-step (f, ss) is key 32f + 16ss + 8(jj>>2) + 4h + (jj&3)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int df = 0; df < 2; ++df) {     // d-fragment innermost: the two output accumulators alternate
            int row0 = 32 * f + 16 * ss + tr_row;
#ifdef ATT_X_NOVREAD     // timing knob: no V^T fragment reads
            half8_t vf = qf[(ss + df) & 3];
#else
            fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4_t*)(vb + v_lds_off(row0, df * 4 + tr_c) + tr_sub));
            fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4_t*)(vb + v_lds_off(row0 + 8, df * 4 + tr_c) + tr_sub));
            half4_t lo4 = __builtin_bit_cast(half4_t, lo), hi4 = __builtin_bit_cast(half4_t, hi);
            half8_t vf = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
#ifndef ATT_X_NOPV
            oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[f][ss], oacc[df], 0, 0, 0);
#else
            oacc[df][(f * 2 + ss) & 15] += (float)vf[0] * (float)pf[f][ss][0];
#endif
          }
      }
    }
#ifndef ATT_X_NOSYNC
#if ATT_NST == 3
    cur = cur == 2 ? 0 : cur + 1;
#else
    cur ^= 1;
#endif
#endif
  }

  // ---- normalise and store: lane owns query row qrow, d = 32*df + 8*(r>>2) + 4*h + (r&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < Sq) {
    half_t* op = out + ((long long)n * Sq + qrow) * ldo + head * 64 + 4 * h;
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

static thread_local int attn_nw_override = 0;     // A/B knobs: waves per workgroup / keys per stage regardless of S
static thread_local int attn_kvb_override = 0;
static thread_local int attn_pipe_mode = 0;       // 0 = by rule, 1 = never the software-pipelined program, 2 = wherever it is legal
extern "C" void lkgd_debug_set_attn_waves(int nw) { attn_nw_override = nw; }
extern "C" void lkgd_debug_set_attn_kvb(int kvb) { attn_kvb_override = kvb; }
extern "C" void lkgd_debug_set_attn_pipe(int mode) { attn_pipe_mode = mode; }

// attn_spatial_pipe.hip: two query tiles per wave, generated software-pipelined main loop; S >= 128
int lkgd_attn_pipe_launch(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                          int32_t ldo, int32_t nbatch, int32_t Sq, int32_t S, int32_t heads, const int32_t* kv_batch_map,
                          float scale, hipStream_t stream);

template <int NW, int KVB>
static int attn_launch(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                       int32_t ldo, int32_t nbatch, int32_t Sq, int32_t S, int32_t heads, const int32_t* kv_batch_map, float scale,
                       hipStream_t stream) {
  constexpr int LDS = ATT_NST * 2 * KVB * 128;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)attn_spatial_kernel<NW, KVB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const int QBLK = NW * 32;
  const int nqb = (Sq + QBLK - 1) / QBLK;
  const long long nwg = (long long)nqb * nbatch * heads;
  if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL((attn_spatial_kernel<NW, KVB>), dim3((unsigned)nwg), dim3(NW * 64), LDS, stream, (const half_t*)q, ldq,
                     (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, Sq, S, heads, kv_batch_map,
                     scale * 1.4426950408889634f, nqb, (int)nwg);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_attn_spatial_qk(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                    void* out, int32_t ldo, int32_t nbatch, int32_t Sq, int32_t S, int32_t heads,
                                    const int32_t* kv_batch_map, float scale, lkgd_stream_t stream) {
  if (!q || !k || !v || !out) return LKGD_E_NULL;
  if (nbatch <= 0 || S <= 0 || Sq <= 0 || heads <= 0) return LKGD_E_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return LKGD_E_ALIGN;
  if (ldq < heads * 64 || ldk < heads * 64 || ldv < heads * 64 || ldo < heads * 64) return LKGD_E_SHAPE;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || ((uintptr_t)out & 7)) return LKGD_E_ALIGN;
  // the software-pipelined program where a workgroup's 512 queries tile the sequence well and the key loop is long enough
  // to amortise its prologue: the 72x128 level (2.74 vs 3.13 ms); at 36x64 (4.5 workgroups per image and head) the two
  // programs are equal, 0.42 ms (tools/attn_bench.py with ATTN_PIPE = 1 / 2, profiles/r04_attn_pipe_opts.txt)
  // (S % 128 != 0 runs the masked form of the statement: CogVideoX's 17 776 tokens)
  if (attn_pipe_mode != 1 && S >= 128 && !attn_nw_override && !attn_kvb_override &&
      (attn_pipe_mode == 2 || (S >= 4096 && Sq >= 4096)))
    return lkgd_attn_pipe_launch(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, S, heads, kv_batch_map, scale,
                                 (hipStream_t)stream);
  // queries per workgroup: 512 at S >= 8192, 256 at S >= 2304, else 128 (tools/attn_bench.py with ATTN_WAVES)
  const int nw = attn_nw_override ? attn_nw_override : (Sq >= 8192 ? 16 : Sq >= 2304 ? 8 : 4);
  const int kvb = attn_kvb_override ? attn_kvb_override : (nw == 16 ? ATT_KVB16 : 64);
  hipStream_t st = (hipStream_t)stream;
#define ATT_ARGS q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, S, heads, kv_batch_map, scale, st
  if (nw == 16) return kvb == 128 ? attn_launch<16, 128>(ATT_ARGS) : attn_launch<16, 64>(ATT_ARGS);
  if (nw == 8) return kvb == 128 ? attn_launch<8, 128>(ATT_ARGS) : attn_launch<8, 64>(ATT_ARGS);
  return attn_launch<4, 64>(ATT_ARGS);
#undef ATT_ARGS
}

extern "C" int lkgd_attn_spatial(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                 void* out, int32_t ldo, int32_t nbatch, int32_t S, int32_t heads,
                                 const int32_t* kv_batch_map, float scale, lkgd_stream_t stream) {
  return lkgd_attn_spatial_qk(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, S, S, heads, kv_batch_map, scale, stream);
}
