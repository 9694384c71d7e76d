// Persistent "streaming" MFMA GEMM / implicit-conv kernel (include/lkgd_hip.h section 1) - the main kernel.
//
//   * one 512-thread workgroup per CU (8 waves as 4(M) x 2(N), 64x64 outputs per wave), 256x128 output tiles, BK = 64;
//   * every workgroup owns a contiguous range of output tiles (n fastest) and walks ALL their K-tiles as ONE stream
//     through a 3 x 48 KiB LDS ring filled by LDS-DMA: two K-tiles are always in flight, also ACROSS output tiles, so the
//     HBM latency of a tile's first K-steps and its whole epilogue are overlapped with the previous tile's work;
//   * operands are swapped in the MFMA (A = weight rows, B = token rows): the accumulator then has the TOKEN on the lane
//     and 4 consecutive OUTPUT CHANNELS in consecutive registers, so the epilogue (bias, row-indexed bias, GEGLU, two
//     scaled residuals) runs straight out of registers with 8-byte row-contiguous loads/stores - no LDS staging, no
//     barrier, the ring is never disturbed;
//   * GEGLU: packed weight rows interleave 32 hidden | 32 gate per wave, so both factors of an output sit in the same
//     lane and register index of the wave's two n-fragments.
#include "gemm_common.h"

#define SBM 256
#define SBN 128
#define SNT 512
#define SSTAGE_BYTES ((SBM + SBN) * BK * 2)   // 48 KiB
#define SNSTAGE 3
#define SLDS (SNSTAGE * SSTAGE_BYTES)         // 144 KiB

__device__ __forceinline__ float gelu_fast(float x) {
  // exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below fp16 resolution)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erf(|x|/sqrt2)
  const float erf_s = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erf_s);
}

struct TileIter {        // a workgroup's walk over its tiles
  int tile, tile_end;    // current / one-past-last linear tile id
  int tiles_n;
};

__global__ __launch_bounds__(SNT, 2) void lkgd_gemm_stream_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int h = lane >> 5, l31 = lane & 31;

  // ---- tile schedule.  Blocks b and b+8 share an XCD (round-robin dispatch).  Each XCD gets a contiguous range of
  //      tiles (n fastest); inside it the XCD's workgroups take tiles ROUND-ROBIN (c, c+nc, c+2nc, ...), so at any time
  //      the ~32 CUs of one XCD work on ~32 adjacent tiles: they share the token (A) tile and the weight tiles through
  //      that XCD's L2 instead of each CU dragging private tiles through the (slower) fabric.
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  const int nc = (G - xcd + 7) >> 3;                       // workgroups on this XCD label
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);         // this XCD's tile range [xb, xe)
  const int xe = xb + q8 + (xcd < r8 ? 1 : 0);
  const int my_tiles = (xe - xb - c + nc - 1) / nc;        // tiles xb+c, xb+c+nc, ... < xe
  const int tile_begin = xb + c;
  const int nk = p.K / BK;
  const int total = (my_tiles > 0 ? my_tiles : 0) * nk;    // K-tiles in this workgroup's stream
  if (total <= 0) return;

  // ---- staging side state (runs two K-tiles ahead of the compute side)
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  AGather<4> ag;
  const half_t* brow[2];
  int st_tile = tile_begin - nc, st_kt = nk;   // forces a tile setup on the first stage()
  auto stage = [&](int buf) {
    if (st_kt == nk) {            // next output tile: re-derive this thread's gather rows
      st_kt = 0;
      st_tile += nc;
      const int tm = st_tile / tiles_n, tn = st_tile - tm * tiles_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) ag.row[i] = a_row(p, tm * SBM + srow + 64 * i);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int n = tn * SBN + srow + 64 * i;
        brow[i] = n < p.N ? (const half_t*)p.w + (long long)n * p.K + schunk * 8 : nullptr;
      }
      a_segment<4>(p, ag, 0, schunk);
    }
    const int k0 = st_kt * BK;
    if (k0 >= ag.seg_end) a_segment<4>(p, ag, k0, schunk);
    char* sa = smem + buf * SSTAGE_BYTES;
    char* sb = sa + SBM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_chunk<4>(ag, i, k0), sa + (w * 64 + 512 * i) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(brow[i] ? brow[i] + k0 : (const half_t*)p.zeros, sb + (w * 64 + 512 * i) * 16);
    ++st_kt;
  };

  // acc[i][j]: n-fragment i (weight rows), m-fragment j (token rows); lane = token, registers = channels
  float16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int x_off[2], w_off[2], x_sw[2], w_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int rx = wr * 64 + i * 32 + l31;       // token row inside the tile
    int rw = wc * 64 + i * 32 + l31;       // weight row inside the tile
    x_off[i] = rx * 128; x_sw[i] = (rx >> 1) & 7;
    w_off[i] = SBM * BK * 2 + rw * 128; w_sw[i] = (rw >> 1) & 7;
  }

  stage(0);
  if (total > 1) stage(1);
  int cur = 0, kt = 0, tile = tile_begin;
  bool skip_wait = false;
  for (int s = 0; s < total; ++s) {
    // K-tile s must have landed.  vmcnt is in issue order, so "at most 6 outstanding" after K-tile s+1 (6 LDS-DMA ops
    // per thread) has been issued retires K-tile s.  When the previous step ended an output tile, this wait was already
    // taken BEFORE that tile's epilogue traffic (see below).
    if (!skip_wait) {
      if (s + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    skip_wait = false;
    __builtin_amdgcn_s_barrier();
    if (s + 2 < total) {
      int nb = cur + 2; if (nb >= SNSTAGE) nb -= SNSTAGE;
      stage(nb);
    }
    const char* sbase = smem + cur * SSTAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8_t xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xf[i] = *(const half8_t*)(sbase + x_off[i] + (((ks * 2 + h) ^ x_sw[i]) << 4));
        wf[i] = *(const half8_t*)(sbase + w_off[i] + (((ks * 2 + h) ^ w_sw[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == SNSTAGE ? 0 : cur + 1;

    if (++kt == nk) {
      // ------------------------------------------------------------------ epilogue of `tile`, straight from registers
      // take step s+1's ring wait now, while the only outstanding ops are K-tiles s+1 and s+2: the epilogue's own loads
      // and stores then never sit between a K-tile and its wait
      if (s + 1 < total) {
        if (s + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        skip_wait = true;
      }
      kt = 0;
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      tile += nc;
      const int m0 = tm * SBM + wr * 64, n0 = tn * SBN + wc * 64;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      if (!p.geglu) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long m = m0 + j * 32 + l31;
          if (m < p.M) {
            long long idx = 0;
            if (rbp) idx = ((m / p.rb_d1) * p.rb_m1 + (m % p.rb_d2) + p.rb_c0) % p.rb_md;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int n = n0 + i * 32 + 8 * g + 4 * h;
                if (n < p.N) {
                  float4_t v;
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
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
        // wave columns [0,32) = hidden, [32,64) = gate of output columns tn*64 + wc*32 + [0,32)
        const int oc0 = tn * 64 + wc * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long m = m0 + j * 32 + l31;
          if (m < p.M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int c = 8 * g + 4 * h;
              float4_t bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
              if (p.bias) {
                bh = *(const float4_t*)(p.bias + n0 + c);
                bg = *(const float4_t*)(p.bias + n0 + 32 + c);
              }
              half4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                o[e] = (half_t)((acc[0][j][4 * g + e] + bh[e]) * gelu_fast(acc[1][j][4 * g + e] + bg[e]));
              *(half4_t*)(outp + m * p.ldc + oc0 + c) = o;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  }
}

extern "C" int lkgd_gemm_stream_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lkgd_gemm_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SLDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  int tiles_m = (d->M + SBM - 1) / SBM, tiles_n = (d->N + SBN - 1) / SBN;
  long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  hipLaunchKernelGGL(lkgd_gemm_stream_kernel, dim3(grid), dim3(SNT), SLDS, stream, *d, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
