// Wide-tile persistent MFMA GEMM / implicit-conv kernel: 256 (tokens) x 320 (channels) output tiles.
//
// Why 320: every channel count on the SVD hot path (320, 640, 960, 1280, 1920, 2560, 3840, 5120, 10240) is a multiple of
// 320, so there is no padded-column waste (128-wide tiles lose 17 % at N = 320), and a 256x320 tile moves
// (256+320)*64*2 B per 10.5 MFLOP K-step = 29 B per MFMA-cycle per CU - within what a CU can pull from its XCD's L2,
// where the 256x128 tile of gemm_stream.hip needs 47 B/cycle and is load-bound.
//
//   * 512 threads, 8 waves as 4(M) x 2(N); a wave owns 64 tokens x 160 channels = 4 x 10 fragments of
//     v_mfma_f32_16x16x32_f16 (160 accumulator VGPRs); BK = 64; two LDS stages of 72 KiB (one K-tile, 2560 MFMA-cycles
//     of cover, always in flight);
//   * persistent workgroups, K-tile stream continuous across output tiles, XCD-cooperative tile schedule and
//     swapped-operand register epilogue exactly as gemm_stream.hip;
//   * GEGLU: packed rows interleave 80 hidden | 80 gate per wave (5 + 5 fragments): both factors in the same lane/register.
#include "gemm_common.h"

#define WBM 256
#define WBN 320
#define WNT 512
#define WSTAGE_BYTES ((WBM + WBN) * BK * 2)   // 72 KiB
#define WLDS (2 * WSTAGE_BYTES)               // 144 KiB

__device__ __forceinline__ float gelu_fast_w(float x) {
  // exact-erf GELU, erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}

__global__ __launch_bounds__(WNT, 2) void lkgd_gemm_wide_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- XCD-cooperative persistent schedule (see gemm_stream.hip)
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  const int nc = (G - xcd + 7) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);
  const int xe = xb + q8 + (xcd < r8 ? 1 : 0);
  const int my_tiles = (xe - xb - c + nc - 1) / nc;
  const int tile_begin = xb + c;
  const int nk = p.K / BK;
  const int total = (my_tiles > 0 ? my_tiles : 0) * nk;
  if (total <= 0) return;

  // ---- staging state
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  AGather<4> ag;                 // rows are re-derived per segment (register diet: 160 accumulators live)
  int st_m0 = 0;
  long long wbase = 0;        // element offset of this thread's first weight row chunk; rows srow + 64*i
  int wvalid = 0;             // bit i: weight row srow + 64*i < N
  int st_tile = tile_begin - nc, st_kt = nk;
  auto stage = [&](int buf) {
    if (st_kt == nk) {
      st_kt = 0;
      st_tile += nc;
      const int tm = st_tile / tiles_n, tn = st_tile - tm * tiles_n;
      st_m0 = tm * WBM + srow;
      const int n = tn * WBN + srow;
      wbase = (long long)n * p.K + schunk * 8;
      wvalid = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) wvalid |= (n + 64 * i < p.N ? 1 : 0) << i;
      ag.seg_end = 0;
    }
    const int k0 = st_kt * BK;
    if (k0 >= ag.seg_end) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ag.row[i] = a_row(p, st_m0 + 64 * i);
      a_segment<4>(p, ag, k0, schunk);
    }
    char* sx = smem + buf * WSTAGE_BYTES;
    char* sw = sx + WBM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_chunk<4>(ag, i, k0), sx + (w * 64 + 512 * i) * 16);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const half_t* src = (wvalid >> i) & 1 ? (const half_t*)p.w + wbase + (long long)(64 * i) * p.K + k0
                                            : (const half_t*)p.zeros;
      glds16(src, sw + (w * 64 + 512 * i) * 16);
    }
    ++st_kt;
  };

  // acc[ni][mi]: 16 channels x 16 tokens; lane = token (l15), registers = 4 consecutive channels at 4*lq
  float4_t acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // fragment rows: tokens wr*64 + mi*16 + l15, weights wc*160 + ni*16 + l15.  The swizzle key (row>>1)&7 is the same
  // for every fragment of a lane (16*mi, 16*ni, 64*wr, 160*wc are all 0 mod 16), so every fragment address is
  // base + compile-time offset: two chunk offsets (one per k-step) per lane, nothing else in registers
  const int skey = (l15 >> 1) & 7;
  const int x_base = (wr * 64 + l15) * 128;
  const int w_base = WBM * BK * 2 + (wc * 160 + l15) * 128;
  const int ch0 = ((0 + lq) ^ skey) << 4, ch1 = ((4 + lq) ^ skey) << 4;

  stage(0);
  int cur = 0, kt = 0, tile = tile_begin;
  bool skip_wait = false;
  for (int s = 0; s < total; ++s) {
    // K-tile s must have landed (it is the only LDS-DMA batch in flight at this point)
    if (!skip_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    skip_wait = false;
    __builtin_amdgcn_s_barrier();
    if (s + 1 < total) stage(cur ^ 1);
    const char* sb = smem + cur * WSTAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* sk_ = sb + (ks ? ch1 : ch0);
      half8_t xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *(const half8_t*)(sk_ + x_base + j * 2048);
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        half8_t wf = *(const half8_t*)(sk_ + w_base + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[j], acc[i][j], 0, 0, 0);
      }
    }
    cur ^= 1;

    if (++kt == nk) {
      // ---------------------------------------------------------------- epilogue of `tile`, straight from registers.
      // Take step s+1's ring wait first (only K-tile s+1 is outstanding), so epilogue traffic never sits before it.
      if (s + 1 < total) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        skip_wait = true;
      }
      kt = 0;
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      tile += nc;
      const int m0 = tm * WBM + wr * 64 + l15;
      const int n0 = tn * WBN + wc * 160 + 4 * lq;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      if (!p.geglu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_barrier(0);     // keep one token fragment's loads/stores from piling onto the next
          const long long m = m0 + j * 16;
          if (m < p.M) {
            long long idx = 0;
            if (rbp) idx = ((m / p.rb_d1) * p.rb_m1 + (m % p.rb_d2) + p.rb_c0) % p.rb_md;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
              const int n = n0 + i * 16;
              if (n < p.N) {
                float4_t v = acc[i][j];
                if (p.bias) v += *(const float4_t*)(p.bias + n);
                if (rbp) {
                  half4_t rb = *(const half4_t*)(rbp + idx * p.ldrb + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
                }
                v *= p.s_acc;
                if (r1p) {
                  half4_t r = *(const half4_t*)(r1p + m * p.ldr1 + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] += p.r1 * (float)r[e];
                }
                if (r2p) {
                  half4_t r = *(const half4_t*)(r2p + m * p.ldr2 + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] += p.r2 * (float)r[e];
                }
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
                *(half4_t*)(outp + m * p.ldc + n) = o;
              }
            }
          }
        }
      } else {
        // wave channels [0,80) = hidden, [80,160) = gate of output columns tn*160 + wc*80 + [0,80)
        const int oc0 = tn * 160 + wc * 80 + 4 * lq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          const long long m = m0 + j * 16;
          if (m < p.M) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              float4_t hv = acc[i][j], gv = acc[i + 5][j];
              if (p.bias) {
                hv += *(const float4_t*)(p.bias + n0 + i * 16);
                gv += *(const float4_t*)(p.bias + n0 + 80 + i * 16);
              }
              half4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_fast_w(gv[e]));
              *(half4_t*)(outp + m * p.ldc + oc0 + i * 16) = o;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
}

extern "C" int lkgd_gemm_wide_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lkgd_gemm_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  int tiles_m = (d->M + WBM - 1) / WBM, tiles_n = (d->N + WBN - 1) / WBN;
  long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  hipLaunchKernelGGL(lkgd_gemm_wide_kernel, dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
