// MFMA fp16 GEMM with implicit-convolution A operand and fused epilogue (include/lkgd_hip.h section 1).
//
// Design (gfx950 / CDNA4):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 v_mfma_f32_32x32x16_f16,
//     64 fp32 accumulator VGPRs), BK = 64, two LDS stages of 32 KiB -> 2 workgroups per CU.
//   * both operands are K-contiguous ([M][K] tokens, [N][K] weights), so A and B fragments are one ds_read_b128 each.
//   * staging is LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear, the bank-conflict swizzle
//     (16-byte chunk ^= (row>>1)&7 inside each 128-byte row) is applied to the per-lane SOURCE address and to the
//     fragment read.  The per-lane source address is also what makes the A operand an implicit convolution: every
//     lane computes where "its" 16 bytes live (3x3 tap with halo, stride-2, nearest-2x upsample, frame shift,
//     two-tensor channel concat) and out-of-image lanes read a zero page.
//   * one barrier per K-tile (stage t+1 is issued before the MFMAs of tile t).
//   * epilogue: accumulators -> LDS (fp32, reusing the staging buffers) -> row-contiguous 8-byte stores with bias,
//     row-indexed bias (time embedding / positional / cross-attention), GEGLU, two scaled residuals (AlphaBlender).
//   * blockIdx -> tile map is XCD-aware (each XCD's L2 sees a contiguous range of tiles, n fastest).
#include "gemm_common.h"

#define BM 128
#define BN 128
#define STAGE_BYTES (2 * BM * BK * 2)  // A 16 KiB + B 16 KiB
#define GEMM_LDS (2 * STAGE_BYTES)     // 64 KiB (also holds the fp32 C tile in the epilogue)

// ksplit > 1: blockIdx.y owns K-tiles [y*per, (y+1)*per) and leaves its fp32 partial tile in ws[y][M][N]
__global__ __launch_bounds__(256, 2) void lkgd_gemm_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n, int ksplit,
                                                           int per, float* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 1, wc = w & 1;

  const int nwg = tiles_m * tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread staging rows: thread t fills LDS slots t + 256*i (slot = row*8 + chunk), i = 0..3
  const int srow = t >> 3;                               // + 32*i
  const int schunk = (t & 7) ^ ((t >> 4) & 7);           // logical 16-byte chunk this thread fetches (swizzled)
  AGather<4> ag;
  const half_t* brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ag.row[i] = a_row(p, m0 + srow + 32 * i);
    int n = n0 + srow + 32 * i;
    brow[i] = n < p.N ? (const half_t*)p.w + (long long)n * p.K + schunk * 8 : nullptr;
  }
  const int nk_all = p.K / BK;
  const int kb = ksplit > 1 ? (int)blockIdx.y * per : 0;
  const int ke = ksplit > 1 ? (kb + per < nk_all ? kb + per : nk_all) : nk_all;
  a_segment<4>(p, ag, kb * BK, schunk);

  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + BM * BK * 2;
    const int k0 = kt * BK;
    if (k0 >= ag.seg_end) a_segment<4>(p, ag, k0, schunk);      // wave-uniform: K-tiles are staged in order
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_chunk<4>(ag, i, k0), sa + (w * 64 + 256 * i) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(brow[i] ? brow[i] + k0 : (const half_t*)p.zeros, sb + (w * 64 + 256 * i) * 16);
  };

  float16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses (bytes inside a stage): row*128 + ((ks*2 + h) ^ ((row>>1)&7))*16
  const int h = lane >> 5;
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = wr * 64 + i * 32 + (lane & 31);
    int rb = wc * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
    b_off[i] = BM * BK * 2 + rb * 128; b_sw[i] = (rb >> 1) & 7;
  }

  stage(0, kb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kb; kt < ke; ++kt) {
    const int cur = (kt - kb) & 1;
    if (kt + 1 < ke) stage(cur ^ 1, kt + 1);
    const char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8_t af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *(const half8_t*)(sbase + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
        bf[i] = *(const half8_t*)(sbase + b_off[i] + (((ks * 2 + h) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS fp32 tile [128][128] -> row-contiguous stores
  float* ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        int col = wc * 64 + j * 32 + (lane & 31);
        ct[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();

  if (ksplit > 1) {
    // partial tile -> workspace, 16-byte row-contiguous stores; the epilogue runs in lkgd_gemm_splitk_reduce
    float* dst = ws + (long long)blockIdx.y * p.M * p.N;
    const int col = (t & 31) * 4;
    if (n0 + col < p.N) {
      for (int row = t >> 5; row < BM && m0 + row < p.M; row += 8)
        *(float4_t*)(dst + (long long)(m0 + row) * p.N + n0 + col) = *(const float4_t*)(ct + row * BN + col);
    }
    return;
  }
  gemm_epilogue<BM, BN, 256>(p, ct, t, m0, n0, tn);
}

// ------------------------------------------------------------------------------------------------------------------
// Few-row variant (round 5): the same 128x128 tile and wave layout on a FOUR-stage LDS ring (128 KiB, one workgroup per
// CU), three K-tiles of LDS-DMA in flight behind a counted s_waitcnt vmcnt.  The two-stage kernel above waits for every
// K-tile's loads before its MFMAs (vmcnt(0) + barrier per K-step): with a second workgroup on the CU that latency is
// hidden, but the problems of a frame-sharded rank (576 - 9216 rows) give a CU ONE workgroup or none, and a K-step then
// costs the whole L2 / fabric round trip - 1.3-1.8 us per K-tile measured (profiles/r05_plan_profile_base.txt: 2304 x
// 1280 x 1280 in 36 us unsplit).  ksplit > 1 as above: blockIdx.y owns K-tiles [y*per, (y+1)*per).
#define MID_NST 4
#define MID_NT 512
#define MID_LDS (MID_NST * STAGE_BYTES)   // 128 KiB (holds the 64 KiB fp32 C tile in the epilogue)
#define MID_LOADS 4                       // LDS-DMA instructions per thread and stage (2 A + 2 B)

__global__ __launch_bounds__(MID_NT, 2) void lkgd_gemm_mid_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n, int ksplit,
                                                                  int per, float* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;              // 0..7: two waves per SIMD, so one wave's fragment reads and LDS-DMA issue run under
  const int wr = w >> 1, wc = w & 1; // the other's MFMAs; wave tile 32 x 64 (4 x 2 waves)

  const int nwg = tiles_m * tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: slot = row*8 + chunk; thread t fills A slots t + 512*i and B slots t + 512*i (i < 2): row = (t>>3) + 64*i
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);           // logical 16-byte chunk this thread fetches (swizzled)
  AGather<2> ag;
  const half_t* brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ag.row[i] = a_row(p, m0 + srow + 64 * i);
    int n = n0 + srow + 64 * i;
    brow[i] = n < p.N ? (const half_t*)p.w + (long long)n * p.K + schunk * 8 : nullptr;
  }
  const int nk_all = p.K / BK;
  const int kb = ksplit > 1 ? (int)blockIdx.y * per : 0;
  const int ke = ksplit > 1 ? (kb + per < nk_all ? kb + per : nk_all) : nk_all;
  const int nkt = ke - kb;
  a_segment<2>(p, ag, kb * BK, schunk);

  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + BM * BK * 2;
    const int k0 = kt * BK;
    if (k0 >= ag.seg_end) a_segment<2>(p, ag, k0, schunk);      // wave-uniform: K-tiles are staged in order
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_chunk<2>(ag, i, k0), sa + (w * 64 + 512 * i) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(brow[i] ? brow[i] + k0 : (const half_t*)p.zeros, sb + (w * 64 + 512 * i) * 16);
  };

  float16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // fragment read addresses (bytes inside a stage): row*128 + ((ks*2 + h) ^ ((row>>1)&7))*16
  const int h = lane >> 5;
  const int ra = wr * 32 + (lane & 31);
  const int a_off = ra * 128, a_sw = (ra >> 1) & 7;
  int b_off[2], b_sw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int rb = wc * 64 + j * 32 + (lane & 31);
    b_off[j] = BM * BK * 2 + rb * 128; b_sw[j] = (rb >> 1) & 7;
  }

  for (int s = 0; s < MID_NST - 1 && s < nkt; ++s) stage(s, kb + s);
  int cur = 0;
  for (int i = 0; i < nkt; ++i) {
    // tile i must have landed: this thread leaves only the (at most two) NEWER tiles' loads outstanding; the barrier then
    // makes every thread's tile-i loads visible and frees the buffer of tile i-1 for restaging
    const int newer = nkt - 1 - i;
#ifndef MID_X_NOBAR        /* MID_X_*: timing experiments only (tools/micro/mid_knobs.sh), results wrong by construction */
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    const char* sbase = smem + cur * STAGE_BYTES;
    half8_t af[4], bf[4][2];                     // the whole K-tile's fragments: 12 reads in flight in front of the MFMAs
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#ifdef MID_X_NOLDS
      af[ks] = (half8_t)(half_t)(float)(lane + i);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[ks][j] = (half8_t)(half_t)(float)(lane - i + j);
#else
      af[ks] = *(const half8_t*)(sbase + a_off + (((ks * 2 + h) ^ a_sw) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[ks][j] = *(const half8_t*)(sbase + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
#endif
    }
#ifndef MID_X_NODMA
    if (i + MID_NST - 1 < nkt) {
      int nb = cur + MID_NST - 1; if (nb >= MID_NST) nb -= MID_NST;
      stage(nb, kb + i + MID_NST - 1);
    }
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#ifdef MID_X_NOMFMA
        acc[j][ks] += (float)af[ks][j] * (float)bf[ks][j][0];
#else
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bf[ks][j], acc[j], 0, 0, 0);
#endif
      }
    cur = cur + 1 == MID_NST ? 0 : cur + 1;
  }
  __syncthreads();   // all waves done reading the ring before it becomes the C tile

  float* ct = (float*)smem;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      int col = wc * 64 + j * 32 + (lane & 31);
      ct[row * BN + col] = acc[j][r];
    }
  __syncthreads();

  if (ksplit > 1) {
    float* dst = ws + (long long)blockIdx.y * p.M * p.N;
    const int col = (t & 31) * 4;
    if (n0 + col < p.N) {
      for (int row = t >> 5; row < BM && m0 + row < p.M; row += 16)
        *(float4_t*)(dst + (long long)(m0 + row) * p.N + n0 + col) = *(const float4_t*)(ct + row * BN + col);
    }
    return;
  }
  gemm_epilogue<BM, BN, MID_NT>(p, ct, t, m0, n0, tn);
}

// second pass of a split-K GEMM: 32 x 128 output tile per workgroup, partials added in slice order, then the shared epilogue
__global__ __launch_bounds__(256) void lkgd_gemm_splitk_reduce(const lkgd_gemm_desc p, int tiles_n, int ksplit,
                                                               const float* ws) {
  __shared__ __attribute__((aligned(16))) float ct[32 * BN];
  const int t = threadIdx.x;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 32, n0 = tn * BN;
  const int col = (t & 31) * 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (t >> 5) + 8 * it;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    if (m0 + row < p.M && n0 + col < p.N) {
      const float* src = ws + (long long)(m0 + row) * p.N + n0 + col;
      for (int s = 0; s < ksplit; ++s) v += *(const float4_t*)(src + (long long)s * p.M * p.N);
    }
    *(float4_t*)(ct + row * BN + col) = v;
  }
  __syncthreads();
  gemm_epilogue<32, BN, 256>(p, ct, t, m0, n0, tn);
}

// ------------------------------------------------------------------------------------------------------------------
// Large-M variant: 256x128 tile, 512 threads (8 waves as 4x2, 64x64 per wave), BK = 64, THREE LDS stages of 48 KiB
// (144 KiB, one workgroup per CU, two waves per SIMD).  Two K-tiles of LDS-DMA stay in flight across the (single,
// raw) barrier of each K-step behind a COUNTED s_waitcnt vmcnt - the HBM latency that the two-stage kernel above
// exposes every K-step is covered by two K-steps of MFMA work.
#define BM2 256
#define NT2 512
#define STAGE2_BYTES ((BM2 + BN) * BK * 2)   // A 32 KiB + B 16 KiB
#define NSTAGE2 3
#define GEMM2_LDS (NSTAGE2 * STAGE2_BYTES)   // 144 KiB (holds the 128 KiB fp32 C tile in the epilogue)

__global__ __launch_bounds__(NT2, 2) void lkgd_gemm_kernel_256(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;              // 0..7
  const int wr = w >> 1, wc = w & 1;
  const int nwg = tiles_m * tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM2, n0 = tn * BN;

  // staging: A slots t + 512*i (i<4): row = (t>>3) + 64*i; B slots t + 512*i (i<2)
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  AGather<4> ag;
  const half_t* brow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) ag.row[i] = a_row(p, m0 + srow + 64 * i);
  a_segment<4>(p, ag, 0, schunk);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int n = n0 + srow + 64 * i;
    brow[i] = n < p.N ? (const half_t*)p.w + (long long)n * p.K + schunk * 8 : nullptr;
  }
  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * STAGE2_BYTES;
    char* sb = sa + BM2 * BK * 2;
    const int k0 = kt * BK;
    if (k0 >= ag.seg_end) a_segment<4>(p, ag, k0, schunk);      // wave-uniform: K-tiles are staged in order
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_chunk<4>(ag, i, k0), sa + (w * 64 + 512 * i) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(brow[i] ? brow[i] + k0 : (const half_t*)p.zeros, sb + (w * 64 + 512 * i) * 16);
  };

  float16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int h = lane >> 5;
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = wr * 64 + i * 32 + (lane & 31);
    int rb = wc * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
    b_off[i] = BM2 * BK * 2 + rb * 128; b_sw[i] = (rb >> 1) & 7;
  }

  const int nk = p.K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed: this thread leaves only the NEWER tile's 6 LDS-DMA ops outstanding, then the
    // barrier makes every thread's tile-kt loads visible and frees buffer (kt-1)%3 for restaging
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      int nb = cur + 2; if (nb >= NSTAGE2) nb -= NSTAGE2;
      stage(nb, kt + 2);
    }
    const char* sbase = smem + cur * STAGE2_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8_t af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *(const half8_t*)(sbase + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
        bf[i] = *(const half8_t*)(sbase + b_off[i] + (((ks * 2 + h) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NSTAGE2 ? 0 : cur + 1;
  }
  __syncthreads();   // all waves done reading the ring before it becomes the C tile

  float* ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        int col = wc * 64 + j * 32 + (lane & 31);
        ct[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();
  gemm_epilogue<BM2, BN, NT2>(p, ct, t, m0, n0, tn);
}


static int check_desc(const lkgd_gemm_desc* d) {
  if (!d || !d->a0 || !d->w || !d->out || !d->zeros) return LKGD_E_NULL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return LKGD_E_SHAPE;
  if (d->K % BK) return LKGD_E_SHAPE;
  if (d->N % 4) return LKGD_E_SHAPE;
  if (d->geglu != 0 && d->geglu != 32 && d->geglu != 80) return LKGD_E_MODE;
  if (d->geglu && (d->N % (4 * d->geglu) || d->rowbias || d->res1 || d->res2)) return LKGD_E_SHAPE;
  if (d->ldc % 4 || d->lda0 % 8) return LKGD_E_ALIGN;
  if (!aligned16(d->a0) || !aligned16(d->w) || !aligned16(d->zeros) || ((uintptr_t)d->out & 7)) return LKGD_E_ALIGN;
  if (d->a1 && (!aligned16(d->a1) || d->lda1 % 8)) return LKGD_E_ALIGN;
  if (d->bias && !aligned16(d->bias)) return LKGD_E_ALIGN;
  if (d->rowbias && (d->ldrb % 4 || d->rb_d1 <= 0 || d->rb_d2 <= 0 || d->rb_md <= 0 || d->rb_c0 < 0)) return LKGD_E_SHAPE;
  if (d->res1 && d->ldr1 % 4) return LKGD_E_ALIGN;
  if (d->res2 && d->ldr2 % 4) return LKGD_E_ALIGN;
  switch (d->mode) {
    case LKGD_A_PLAIN:
      if (d->csplit % 8 || d->csplit <= 0) return LKGD_E_SHAPE;
      if (d->csplit < d->K && !d->a1) return LKGD_E_NULL;
      if (d->csplit < d->K && d->csplit % BK) return LKGD_E_SHAPE;
      break;
    case LKGD_A_CONV3X3:
      if (d->Cin <= 0 || d->Cin % BK || d->K != 9 * d->Cin) return LKGD_E_SHAPE;
      if (d->csplit <= 0 || d->csplit % BK) return LKGD_E_SHAPE;
      if (d->csplit < d->Cin && !d->a1) return LKGD_E_NULL;
      if (d->Hout <= 0 || d->Wout <= 0 || d->Hin <= 0 || d->Win <= 0) return LKGD_E_SHAPE;
      if (d->stride != 1 && d->stride != 2) return LKGD_E_SHAPE;
      if (d->ups != 0 && d->ups != 1) return LKGD_E_SHAPE;
      if (d->M % (d->Hout * d->Wout)) return LKGD_E_SHAPE;
      if (d->pad_off != 0 && d->pad_off != 1) return LKGD_E_SHAPE;
      // output size must match: pad 1 on every side: floor((Hv + 2 - 3)/stride) + 1;  pad_off (0 top/left, 1 bottom/right):
      // floor((Hv + 1 - 3)/stride) + 1
      if (d->Hout != (((d->Hin << d->ups) - 1 - d->pad_off) / d->stride) + 1) return LKGD_E_SHAPE;
      if (d->Wout != (((d->Win << d->ups) - 1 - d->pad_off) / d->stride) + 1) return LKGD_E_SHAPE;
      break;
    case LKGD_A_TCONV3:
      if (d->Cin <= 0 || d->Cin % BK || d->K != 3 * d->Cin) return LKGD_E_SHAPE;
      if (d->F <= 0 || d->HW <= 0 || d->Floc <= 0 || d->f_off < 0 || d->f_off + d->Floc > d->F) return LKGD_E_SHAPE;
      if (d->M % (d->Floc * d->HW)) return LKGD_E_SHAPE;
      break;
    case LKGD_A_CONV3X3_C8:
      if (d->Cin != 8 || d->K != 128 || d->lda0 != 8 || d->stride != 1 || d->ups != 0 || d->pad_off != 0) return LKGD_E_SHAPE;
      if (d->Hout != d->Hin || d->Wout != d->Win || d->M % (d->Hout * d->Wout)) return LKGD_E_SHAPE;
      break;
    default:
      return LKGD_E_MODE;
  }
  return LKGD_OK;
}

extern "C" int lkgd_gemm_stream_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus);   // gemm_stream.hip
extern "C" int lkgd_gemm_wide_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus, int ksplit, int wn, int wm);   // gemm_wide.hip
extern "C" int lkgd_gemm_wide_tile_n(int N);     // 320, or 256 for N = 256, 512, 768 ... (the 256x256 form of that program)
extern "C" int lkgd_debug_wide_tile_n_forced();  // the A/B knob's value (0 = rules apply)
extern "C" int lkgd_debug_wide_tile_m_forced();
extern "C" int lkgd_gemm_rowpanel_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus); // gemm_rowpanel.hip
extern "C" int lkgd_gemm_resw_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus);     // gemm_resw.hip
extern "C" int lkgd_gemm_resw_ok(const lkgd_gemm_desc* d, int cus);
extern "C" int lkgd_gemm_resw_colstats_ok(const lkgd_gemm_desc* d);

// tuning/testing knob (not part of the reference-facing ABI): 0 = auto, 1 = force 128x128, 2 = force 256x128 ring,
// 3 = force the persistent streaming kernel (256x128), 4 = force the wide persistent kernel (256x320),
// 5 = force the register-resident row-panel kernel where it applies (plain A, K <= 320),
// 6 = force the resident-weight kernel where it applies (plain A, K <= 320, N % 160 == 0)
static thread_local int gemm_variant_override = 0;
extern "C" void lkgd_debug_set_gemm_variant(int v) { gemm_variant_override = (v >= 1 && v <= 7) ? v : 0; }   // 7 = 128x128 on the four-stage ring
extern "C" void lkgd_debug_set_gemm_splitk(int on);

// Split-K for the 256x320 kernel on problems whose tiles leave CUs idle (fewer tiles than CUs): EQUAL K slices (a divisor
// c <= 8 of K / 64, slices of >= 10 K-tiles, K >= 48 K-tiles in all, partials within the caller's workspace) as virtual
// tiles.  Of the legal c the one whose virtual tiles best fill the CU rounds they occupy wins (ties: fewer slices), if that
// fill is >= 70 % and at least 15 points above the unsplit fill; 0 = none.  Measured per slice count
// (tools/micro/ksplit_sweep.py, profiles/r02_gemm_ksplit_sweep.txt): 64 tiles (4032 x 1280) want 4 slices - 3x3 conv 0.272
// unsplit / 0.122 at 3 / 0.114 at 4, temporal conv 0.104 / 0.062 / 0.060; 128 tiles (8064 rows, the 18x32 level of a
// CFG-parallel rank) want 2 - 3x3 conv 0.274 -> 0.200, FF-out 0.135 -> 0.108; 36 tiles (a rank of 8) want 5-6 even at
// 10-K-tile slices (temporal conv 0.046 vs 0.050 on 128x128 tiles); below 48 K-tiles the fp32 partials cost more than the idle
// CUs (9216 x 640 x 1920: 0.046 split vs 0.039 on 128x128 tiles).
static thread_local bool gemm_splitk_enabled = true;
static thread_local int wide_ksplit_forced = 0;      // tools/micro/ksplit_sweep.py: force this slice count where it is legal
extern "C" void lkgd_debug_set_wide_ksplit(int k) { wide_ksplit_forced = k < 0 ? 0 : k; }
static int wide_split(const lkgd_gemm_desc* d, long long tiles_wide, int cus) {
  if (d->geglu || !d->workspace || !aligned16(d->workspace)) return 0;
  const int nk = d->K / BK;
  if (wide_ksplit_forced) {
    const int c = wide_ksplit_forced;
    return (c >= 2 && nk % c == 0 && (long long)c * d->M * d->N * 4 <= d->workspace_bytes) ? c : 0;
  }
  if (!gemm_splitk_enabled || tiles_wide >= cus || nk < 48) return 0;
  const double unsplit = (double)tiles_wide / cus;
  int best = 0;
  double best_fill = 0.0;
  // measured tile counts: 36, 64, 128 (slices 5-6, 4, 2).  Elsewhere - e.g. ~200 tiles, where the fill rule would pick 5 - the
  // partial traffic c * M * N * 4 bytes is kept within what the measured cases moved (4 slices at 64 tiles = 66 MB x 4)
  for (int c = 2; c <= 8 && c * 10 <= nk; ++c) {
    if (nk % c || (long long)c * d->M * d->N * 4 > d->workspace_bytes) continue;
    if (tiles_wide > 128 && c > 2) continue;
    const long long vt = tiles_wide * c, rounds = (vt + cus - 1) / cus;
    const double fill = (double)vt / (double)(rounds * cus);
    if (fill > best_fill + 1e-9) { best_fill = fill; best = c; }
  }
  return (best_fill >= 0.70 && best_fill >= unsplit + 0.15) ? best : 0;
}

extern "C" void lkgd_debug_set_gemm_splitk(int on) { gemm_splitk_enabled = on != 0; }

// K slices of the four-stage 128x128 kernel (one workgroup per CU): the slice count with the smallest MODELLED time.  A
// workgroup costs a fixed part (prologue, pipeline fill, epilogue or partial-tile store) plus its K-tiles; slicing adds the
// reduce pass, which reads ks fp32 copies of the output.  Constants from tools/micro/mid_knobs.py (time over K at 2304 x 1280:
// 0.75 us per K-tile, 5-6 us fixed; profiles/r05_gemm_mid_knobs.txt); 1 = unsplit.
static thread_local float mid_tk_us = 0.75f, mid_fix_us = 6.0f, mid_red_us = 5.0f, mid_red_tbs = 3.5f;
static thread_local int mid_ksplit_forced = 0;
extern "C" void lkgd_debug_set_mid_model(float tk, float fix, float red, float tbs, int forced) {
  if (tk > 0) mid_tk_us = tk;
  if (fix > 0) mid_fix_us = fix;
  if (red > 0) mid_red_us = red;
  if (tbs > 0) mid_red_tbs = tbs;
  mid_ksplit_forced = forced < 0 ? 0 : forced;
}
static int mid_split(const lkgd_gemm_desc* d, long long nwg, int nk, int cus) {
  if (!d->workspace || !aligned16(d->workspace) || d->N % 4) return 1;
  const long long fit = d->workspace_bytes / ((long long)d->M * d->N * 4);
  if (mid_ksplit_forced) return (mid_ksplit_forced <= nk && mid_ksplit_forced <= fit) ? mid_ksplit_forced : 1;
  if (!gemm_splitk_enabled || nk < 16) return 1;
  int best = 1;
  float best_t = 1e30f;
  for (int ks = 1; ks <= 16 && ks <= fit && ks * 8 <= nk; ++ks) {
    const int per = (nk + ks - 1) / ks;
    const long long blocks = nwg * ((nk + per - 1) / per), rounds = (blocks + cus - 1) / cus;
    float t = (float)rounds * (mid_fix_us + per * mid_tk_us);
    if (ks > 1) t += mid_red_us + (float)ks * (float)d->M * (float)d->N * 4.0f / (mid_red_tbs * 1e6f);
    if (t < best_t * 0.97f) { best_t = t; best = ks; }      // more slices only for a clear gain
  }
  return best;
}

static int gemm_cus(int* cus_out) {
  static std::atomic<int> cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return LKGD_E_LAUNCH;
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LKGD_E_LAUNCH;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  *cus_out = cus;
  return LKGD_OK;
}

// tile-program choice for a checked descriptor: 1 = 128x128, 2 = 256x128 ring, 3 = stream, 4 = wide (256x320), 5 = row-panel,
// 6 = resident-weight; *wide_ks_out = K slices of the 256x320 program (1 = none); negative = LKGD_E_*
static int gemm_pick(const lkgd_gemm_desc* d, int cus, int* wide_ks_out) {   // (7 = 128x128 on the four-stage ring)
  // GEGLU weights are packed for one tile family (interleave width 80 -> 256x320 tiles, 32 -> 128-wide tiles)
  // the persistent kernels move epilogue rows as 16-byte chunks: 8-channel granularity and 16-byte aligned rows
  const bool rows16 = d->N % 8 == 0 && d->ldc % 8 == 0 && aligned16(d->out) &&
                      (!d->res1 || (d->ldr1 % 8 == 0 && aligned16(d->res1))) &&
                      (!d->res2 || (d->ldr2 % 8 == 0 && aligned16(d->res2)));
  // ---- tile-program choice.  Rules are the per-shape winners of the interleaved A/B runs (tools/gemm_shapes_bench.py ab,
  //      profiles/r01_gemm_shapes_ab*.txt), expressed through the quantities that explain them: how many workgroup
  //      slots a tiling fills (a 256-row tiling of M = 16 128 x N = 640 is 126 tiles for 256 CUs), K depth (amortises
  //      the epilogue and the pipeline fill), and N granularity (320-wide tiles for N = 320 / 640 / 1280).
  const int v = gemm_variant_override;
  const bool plain = d->mode == LKGD_A_PLAIN;
  const bool rp_ok = rows16 && plain && d->K <= 320 && d->csplit >= d->K && d->geglu != 80;
  // (the GEGLU epilogue of the 256x320 kernel is compiled into its plain-linear instantiation only)
  const bool wide_ok = (d->geglu == 0 || (d->geglu == 80 && plain)) && d->mode != LKGD_A_CONV3X3_C8 &&
                       d->M < (1 << 24);
  const bool stream_ok = rows16 && d->geglu != 80;
  // resident-weight kernel: a 160-row slab of W[N][K] in LDS per workgroup, every XCD runs all N / 160 slabs
  const bool rw_ok = lkgd_gemm_resw_ok(d, cus) != 0;
  const int wide_n = lkgd_gemm_wide_tile_n(d->N);
  const long long tiles_wide = (long long)((d->M + 255) / 256) * ((d->N + wide_n - 1) / wide_n);
  // N tiles by 320 (by 256 where that divides N and 320 does not: the VAE decoder's 256 / 512) with at most a fifth of all
  // tile columns idle
  const bool n320 = d->N % wide_n == 0 || (long long)((d->N + 319) / 320) * 320 * 4 <= (long long)d->N * 5;
  int pick = 0;   // 1 = 128x128, 2 = 256x128 ring, 3 = stream, 4 = wide (256x320), 5 = rowpanel
  int wide_ks = 1;
  if (d->geglu == 80) {
    if (!wide_ok) return LKGD_E_SHAPE;
    // 80-wide GEGLU interleave: the 256x320 kernel, or - at K <= 320 from 32k rows - the resident-weight kernel (a slab = one
    // wave's 160 packed rows): its waves run their erf epilogues beside each other's K-loops (72x128 level 0.485 vs 0.533 ms,
    // a CFG-parallel rank's 0.246 vs 0.274; profiles/r03_resw_ab.txt)
    pick = (rw_ok && (v == 6 || (v == 0 && d->M >= 32768))) ? 6 : 4;
  } else if (v != 0) {
    pick = v;
  } else if (rw_ok && d->res1 && d->M >= 32768) {
    // K <= 320 projections that carry a residual (attention out, proj_out at the 72x128 level): 0.125 ms against 0.138
    // (row-panel) / 0.152 (256x320) at 258k rows, 0.053 vs 0.059 at 129k.  Without a residual the 256x320 kernel stays
    // ahead (QKV 0.225 vs 0.264, bias-only 0.089 vs 0.096): there the resident-weight kernel's token loads (64 different
    // rows per instruction) and its eight in-order waves do not hide what they wait for
    pick = 6;
  } else if (rp_ok && d->M >= 4096 && d->K >= 192 && (d->res1 || (d->N != 320 && d->M >= 196608))) {
    // K <= 320 projections at 258k rows: A read exactly once.  The 320 x 320 ones only when they carry a residual (its
    // row-coalesced epilogue wins there; without one the 256x320 kernel is ahead).  Without a residual (QKV) the two
    // programs tie at 258k rows since the 256x320 kernel writes its rows through LDS, and at a sharded rank's row counts
    // (129k / 65k / 37k: 504 / 252 / 144 row panels for 256 CUs) the 256x320 tiles fill the CUs better: 0.118 vs 0.125,
    // 0.039 vs 0.052 ms (profiles/r02_gemm_shapes_ab_shards.txt)
    pick = 5;
  } else if (wide_ok && d->geglu == 0 && d->N % 320 == 0 && d->M >= 1024 && (wide_ks = wide_split(d, tiles_wide, cus)) >= 2) {
    pick = 4;                                            // fewer tiles than CUs, deep K: 256x320 tiles over equal K slices
                                                         // (the 9x16 level; the 18x32 / 36x64 levels of sharded ranks)
  } else if (wide_ok && d->geglu == 0 && n320 && d->K >= 320 && tiles_wide * 2 >= cus - 16) {
    pick = 4;                                            // every N = 320 / 640 / 960 / 1280 / 1920 / 3840 shape of the model
                                                         // whose 256x320 tiles fill at least half the CUs; N = 256 / 512 (the
                                                         // VAE decoder) on the 256x256 form of the same program
                                                         // (tools/micro/vae_shapes_bench.py)
  } else if (d->M < 12288) {
    wide_ks = 1;
    // the 9x16 level (M = 4032; also the 9216-row 36x64 level of a rank of 8): 256-row tilings leave most CUs idle; 128x128
    // at two workgroups per CU fills best, except for the wide-N projections (QKV: 16 x 12 tiles of 256x320)
    pick = (wide_ok && plain && d->geglu == 0 && d->N % 320 == 0 && d->N >= 2560 && d->M >= 2048 && !d->res1) ? 4 : 1;
    // 128x128 tiles that give a CU one workgroup at most (the 18x32 / 9x16 levels of a frame-sharded rank, the full model's
    // 9x16 level): the four-stage ring instead of the two-stage kernel, whose K-steps then wait out every load - 2304 x 1280 x
    // 1280 0.022 vs 0.036 ms, the 576-row temporal conv 0.025 vs 0.033, ff-out 0.029 vs 0.036 (profiles/r05_gemm_mid_ab.txt)
    if (pick == 1 && d->geglu == 0 && (long long)((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN) <= cus) pick = 7;
  } else {
    pick = 3;
  }
  // applicability (forced variants fall back the same way)
  if (pick == 6 && !rw_ok) pick = d->geglu == 80 ? 4 : 1;
  if (pick == 5 && !rp_ok) pick = 1;
  if (d->ln_colsum) {          // LayerNorm fold: only the row-panel program holds whole rows in registers
    if (!rp_ok || d->geglu || d->K % 64 || !aligned16(d->ln_colsum) || !(d->ln_eps > 0.f)) return LKGD_E_SHAPE;
    pick = 5;
  }
  if (pick == 4 && !wide_ok) pick = 3;
  if (pick == 3 && (!stream_ok || d->M <= 256)) pick = (d->K >= 960 && d->M > 256) ? 2 : 1;
  if (d->geglu == 80 && pick != 4 && pick != 6) return LKGD_E_SHAPE;   // 80-wide interleave: 256x320 / resident-weight kernels
  *wide_ks_out = wide_ks;
  return pick;
}

// wide_split() for a forced 256x320 variant follows the same slicing rule as the automatic one
static int gemm_wide_slices(const lkgd_gemm_desc* d, int cus, int pick, int wide_ks) {
  if (pick != 4) return 1;
  if (gemm_variant_override == 4) {
    const int wide_n = lkgd_gemm_wide_tile_n(d->N);
    const long long tiles_wide = (long long)((d->M + 255) / 256) * ((d->N + wide_n - 1) / wide_n);
    wide_ks = wide_n == 320 ? wide_split(d, tiles_wide, cus) : 1;
  }
  return wide_ks >= 2 ? wide_ks : 1;
}

// Tile form of an unsliced 256x320-program launch: tile rows wm (256 | 192: wave tile 64 | 48 rows) x tile columns wn (320 | 256).
// The N-only rule (lkgd_gemm_wide_tile_n) fixes wn where only one width divides N; a plain linear whose channel count BOTH widths
// divide (1280, 3840) may take either.  Among the candidates the one with the smallest MODELLED time wins: rounds over the CUs
// x the cost of one tile of that form relative to 256x320 - 1.00 | 0.85 (256x256) | 0.83 (192x320) | 0.72 (192x256), measured
// on one-round shapes (tools/micro/wide_tile_m.py, profiles/r05_wide_tile_m.txt: 36 864 x 320 x 2880 conv 88.8 -> 75.5 us,
// 16 128 x 640 x 5760 conv 154.7 -> 127.7, 8064 x 1280 x 1280 51.6 -> 43.7 / 42.5 / 37.1) - and a smaller tile must win by 3 %.
// On the full forward every large shape keeps 256x320 (1008 tiles = 4 rounds against 1344 = 6 x 0.83); the smaller forms are
// for the levels of a sharded rank, where 256-row tiles leave CUs idle.
static void gemm_wide_form(const lkgd_gemm_desc* d, int cus, int slices, int* wn_out, int* wm_out) {
  const int wn0 = lkgd_gemm_wide_tile_n(d->N);
  *wn_out = wn0;
  *wm_out = 256;
  if (slices != 1) return;
  const int fn = lkgd_debug_wide_tile_n_forced(), fm = lkgd_debug_wide_tile_m_forced();
  if (fn || fm) {                                  // A/B knobs: no rule
    if (fm) *wm_out = fm;
    return;
  }
  const bool both = wn0 == 320 && d->mode == LKGD_A_PLAIN && !d->geglu && d->N % 320 == 0 && d->N % 256 == 0;   // (GEGLU: 80-wide interleave)
  float best = 0.f;
  for (int pass = 0; pass < 4; ++pass) {           // 256x wn0 first: the default, to which the others are compared
    const int wm = (pass & 1) ? 192 : 256, wn = (pass & 2) ? 256 : wn0;
    if ((pass & 2) && !both) continue;
    if (pass && d->cs_rows > 0 && d->cs_rows % wm) continue;       // column sums: whole tiles per GroupNorm sample
    const long long tiles = (long long)((d->M + wm - 1) / wm) * ((d->N + wn - 1) / wn);
    const long long rounds = (tiles + cus - 1) / cus;
    const float unit = (wn == 256 ? 0.85f : 1.0f) * (wm == 192 ? (wn == 256 ? 0.72f / 0.85f : 0.83f) : 1.0f);
    const float cost = (float)rounds * unit;
    if (pass == 0) best = cost;
    else if (cost < best * 0.97f) { best = cost; *wn_out = wn; *wm_out = wm; }
  }
}

extern "C" int lkgd_gemm_colstats_block(const lkgd_gemm_desc* d) {
  if (!d || check_desc(d) != LKGD_OK || d->geglu || d->N % 8) return 0;
  int cus = 0;
  if (gemm_cus(&cus) != LKGD_OK) return 0;
  int wide_ks = 1;
  const int pick = gemm_pick(d, cus, &wide_ks);
  if (pick == 6) return lkgd_gemm_resw_colstats_ok(d) ? 32 : 0;
  // the 256x320 program sums the columns of the row segments it parks in LDS: whole 320-column tiles, unsliced K
  if (pick == 4 && gemm_wide_slices(d, cus, pick, wide_ks) == 1 && d->ldc % 8 == 0 && aligned16(d->out)) {
    int wn, wm;
    gemm_wide_form(d, cus, 1, &wn, &wm);
    return d->N % wn == 0 ? wm : 0;
  }
  return 0;
}

extern "C" int lkgd_gemm_f16(const lkgd_gemm_desc* d, lkgd_stream_t stream) {
  int rc = check_desc(d);
  if (rc != LKGD_OK) return rc;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)lkgd_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_kernel_256, hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM2_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS) !=
            hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  int cus = 0;
  if ((rc = gemm_cus(&cus)) != LKGD_OK) return rc;
  int wide_ks = 1;
  const int pick = gemm_pick(d, cus, &wide_ks);
  if (pick < 0) return pick;
  if (d->colstats && lkgd_gemm_colstats_block(d) == 0) return LKGD_E_SHAPE;   // the chosen program produces no column sums
  if (pick == 6) return lkgd_gemm_resw_launch(d, (hipStream_t)stream, cus);
  if (pick == 5) return lkgd_gemm_rowpanel_launch(d, (hipStream_t)stream, cus);
  if (pick == 4) {
    const int ks = gemm_wide_slices(d, cus, pick, wide_ks);
    int wn, wm;
    gemm_wide_form(d, cus, ks, &wn, &wm);
    rc = lkgd_gemm_wide_launch(d, (hipStream_t)stream, cus, ks, wn, wm);
    if (rc != LKGD_OK || ks == 1) return rc;
    const int tn128 = (d->N + BN - 1) / BN;
    const unsigned rblocks = (unsigned)(((d->M + 31) / 32) * tn128);
    hipLaunchKernelGGL(lkgd_gemm_splitk_reduce, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, *d, tn128, ks,
                       (const float*)d->workspace);
    return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
  }
  if (pick == 3) return lkgd_gemm_stream_launch(d, (hipStream_t)stream, cus);
  int tiles_n = (d->N + BN - 1) / BN;
  if (pick == 7) {
    const int tiles_m = (d->M + BM - 1) / BM;
    const long long nwg = (long long)tiles_m * tiles_n;
    if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
    const int nk = d->K / BK;
    const int ksplit = mid_split(d, nwg, nk, cus);
    if (ksplit > 1) {
      const int per = (nk + ksplit - 1) / ksplit;
      const int ks = (nk + per - 1) / per;           // every slice non-empty
      hipLaunchKernelGGL(lkgd_gemm_mid_kernel, dim3((unsigned)nwg, (unsigned)ks), dim3(MID_NT), MID_LDS, (hipStream_t)stream, *d,
                         tiles_m, tiles_n, ks, per, (float*)d->workspace);
      const unsigned rblocks = (unsigned)(((d->M + 31) / 32) * tiles_n);
      hipLaunchKernelGGL(lkgd_gemm_splitk_reduce, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, *d, tiles_n, ks,
                         (const float*)d->workspace);
    } else {
      hipLaunchKernelGGL(lkgd_gemm_mid_kernel, dim3((unsigned)nwg), dim3(MID_NT), MID_LDS, (hipStream_t)stream, *d, tiles_m,
                         tiles_n, 1, nk, (float*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
  }
  // deep-K problems (3x3 / temporal convs, K >= 960) take the 256x128 three-stage ring: its two K-tiles in flight hide
  // the HBM latency the two-stage kernel exposes every K-step.  Short-K GEMMs (K = 320/640 projections at 258k rows) are
  // epilogue-bound; they keep 128x128 tiles at two workgroups per CU so one workgroup's epilogue overlaps the other's
  // main loop (measured per shape: tools/gemm_shapes_bench.py, profiles/r01_gemm_shapes.txt).
  if (pick == 2) {
    int tiles_m = (d->M + BM2 - 1) / BM2;
    long long nwg = (long long)tiles_m * tiles_n;
    if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
    hipLaunchKernelGGL(lkgd_gemm_kernel_256, dim3((unsigned)nwg), dim3(NT2), GEMM2_LDS, (hipStream_t)stream, *d,
                       tiles_m, tiles_n);
  } else {
    int tiles_m = (d->M + BM - 1) / BM;
    long long nwg = (long long)tiles_m * tiles_n;
    if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
    // split-K when the tiles leave workgroup slots (2 per CU) idle: the smallest slice count whose blocks fill >= 85 % of
    // the rounds they occupy, slices of >= 4 K-tiles when less than half the slots are filled (few-row problems: the reduce
    // pass is tiny) and of >= 32 K-tiles otherwise (the 3x3 convs of the full model's 9x16 level: 320 tiles on 512 slots
    // -> 3 slices, 0.172 -> 0.150 ms; its K = 3840 temporal convs lose more in the reduce pass than they gain);
    // bounded by the caller's workspace (include/lkgd_hip.h: lkgd_gemm_desc.workspace)
    int ksplit = 1, per = d->K / BK;
    const int nk = d->K / BK;
    const long long slots = 2LL * cus;
    if (gemm_splitk_enabled && d->workspace && aligned16(d->workspace) && d->N % 4 == 0 && nk >= 8 &&
        nwg * 10 < slots * 7) {
      const bool few = nwg * 2 <= slots;
      const int min_slice = few ? 4 : 32;
      long long cap = nk / min_slice;
      if (cap > 16) cap = 16;
      const long long fit = d->workspace_bytes / ((long long)d->M * d->N * 4);
      if (cap > fit) cap = fit;
      long long want = 1;
      for (long long ks = 2; ks <= cap; ++ks) {
        const long long blocks = nwg * ks, rounds = (blocks + slots - 1) / slots;
        if (blocks * 100 >= rounds * slots * 85 || (few && ks == cap)) { want = ks; break; }
      }
      if (few && slots / nwg > want && slots / nwg <= cap) want = slots / nwg;   // fill one whole round
      if (want >= 2) {
        per = (int)((nk + want - 1) / want);
        ksplit = (nk + per - 1) / per;           // every slice non-empty
      }
    }
    if (ksplit > 1) {
      hipLaunchKernelGGL(lkgd_gemm_kernel, dim3((unsigned)nwg, (unsigned)ksplit), dim3(256), GEMM_LDS,
                         (hipStream_t)stream, *d, tiles_m, tiles_n, ksplit, per, (float*)d->workspace);
      const unsigned rblocks = (unsigned)(((d->M + 31) / 32) * tiles_n);
      hipLaunchKernelGGL(lkgd_gemm_splitk_reduce, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, *d, tiles_n, ksplit,
                         (const float*)d->workspace);
    } else {
      hipLaunchKernelGGL(lkgd_gemm_kernel, dim3((unsigned)nwg), dim3(256), GEMM_LDS, (hipStream_t)stream, *d, tiles_m,
                         tiles_n, 1, nk, (float*)nullptr);
    }
  }
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
