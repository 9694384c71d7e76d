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

#ifdef LKGD_GEMM_STAMPS
__device__ unsigned long long lkgd_gemm_stamps[256 * 8];
#define STAMP(var) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; }
#else
#define STAMP(var)
#endif

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
  // Waves w and w+4 share a SIMD.  LDS-DMA issue is slow (~100+ cycles per instruction under load) and occupies only the
  // issuing wave, MFMA only the matrix pipe: waves 4-7 therefore run [MFMA, then stage] while waves 0-3 run
  // [stage, then MFMA] after the same barrier, so each SIMD always has one wave feeding the matrix pipe.
  const bool late_stage = __builtin_amdgcn_readfirstlane(t) >= 256;
#ifdef LKGD_GEMM_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, a_wait = 0, a_stage = 0, a_comp = 0, a_epi = 0, t_begin = 0;
  STAMP(t_begin)
#endif
  for (int s = 0; s < total; ++s) {
    STAMP(ts0)
    // K-tile s must have landed.  vmcnt is in issue order, so "at most 6 outstanding" after K-tile s+1 (6 LDS-DMA ops
    // per thread) has been issued retires K-tile s.  When the previous step ended an output tile, this wait was already
    // taken BEFORE that tile's epilogue traffic (see below).
    if (!skip_wait) {
      if (s + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    skip_wait = false;
    __builtin_amdgcn_s_barrier();
    STAMP(ts1)
    if (!late_stage && s + 2 < total) {
      int nb = cur + 2; if (nb >= SNSTAGE) nb -= SNSTAGE;
      stage(nb);
    }
    STAMP(ts2)
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
    if (late_stage && s + 2 < total) {
      int nb = cur + 2; if (nb >= SNSTAGE) nb -= SNSTAGE;
      stage(nb);
    }
    cur = cur + 1 == SNSTAGE ? 0 : cur + 1;
    STAMP(ts3)
#ifdef LKGD_GEMM_STAMPS
    a_wait += ts1 - ts0; a_stage += ts2 - ts1; a_comp += ts3 - ts2;
#endif

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
      // The ring buffer of the K-tile just consumed stays untouched until stage(s+3), which is issued after the NEXT
      // step's barrier: once every wave is done reading it (the barrier below) it serves as per-wave transpose scratch,
      // so that residual reads and output stores are whole 128-byte rows (8 lanes x 16 B) instead of 8-byte pieces at a
      // 2*ldc stride (8x fewer cache-line touches per instruction).
      __builtin_amdgcn_s_barrier();
      char* scr = smem + (cur == 0 ? SNSTAGE - 1 : cur - 1) * SSTAGE_BYTES + w * 6144;   // 32 rows x 144 B per wave
      const int m0 = tm * SBM + wr * 64, n0 = tn * SBN + wc * 64;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      const int crow = lane >> 3, cchunk = lane & 7;       // coalesced map: 8 lanes per 128-byte row, 8 rows per pass
      if (!p.geglu) {
        const int ncol = n0 + cchunk * 8;                   // first of this lane's 8 channels in the coalesced map
        // (1) issue EVERY epilogue load of this tile before consuming any: one exposed memory round trip per tile.
        //     Residual rows come in coalesced (8 lanes x 16 B per row) for both token fragments.
        uint4 rres[2][4];
        if (r1p) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const long long mr = m0 + j * 32 + crow + 8 * it;
              rres[j][it] = (mr < p.M && ncol < p.N) ? *(const uint4*)(r1p + mr * p.ldr1 + ncol)
                                                     : uint4{0u, 0u, 0u, 0u};
            }
        }
        // (2) bias + row-indexed bias + scale, in place in the accumulators
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long m = m0 + j * 32 + l31;
          long long idx = 0;
          if (rbp && m < p.M) idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) + (unsigned)p.rb_c0) % (unsigned)p.rb_md;   // M is an int32
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = n0 + i * 32 + 8 * g + 4 * h;
              float4_t v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
              if (n < p.N) {
                if (p.bias) v += *(const float4_t*)(p.bias + n);
                if (rbp && m < p.M) {
                  half4_t rb = *(const half4_t*)(rbp + idx * p.ldrb + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
                }
              }
              v *= p.s_acc;
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = v[e];
            }
        }
        // (3) per token fragment: residual(s) through the scratch into accumulator layout, result back out as rows
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long mb = m0 + j * 32;
          if (r1p) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *(uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16) = rres[j][it];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                half4_t r = *(const half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += p.r1 * (float)r[e];
              }
          }
          if (r2p) {   // AlphaBlender only (16 GEMMs per forward): its own round trip
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const long long mr = mb + crow + 8 * it;
              uint4 x = {0u, 0u, 0u, 0u};
              if (mr < p.M && ncol < p.N) x = *(const uint4*)(r2p + mr * p.ldr2 + ncol);
              *(uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16) = x;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                half4_t r = *(const half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += p.r2 * (float)r[e];
              }
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              half4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (half_t)acc[i][j][4 * g + e];
              *(half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2) = o;
            }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long mr = mb + crow + 8 * it;
            const uint4 x = *(const uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16);
            if (mr < p.M && ncol < p.N) *(uint4*)(outp + mr * p.ldc + ncol) = x;
          }
        }
      } else {
        // wave columns [0,32) = hidden, [32,64) = gate of output columns tn*64 + wc*32 + [0,32)
        const int oc0 = tn * 64 + wc * 32;
        const int grow = lane >> 2, gchunk = lane & 3;      // 4 lanes per 64-byte row, 16 rows per pass
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long long mb = m0 + j * 32;
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
            *(half4_t*)(scr + l31 * 144 + c * 2) = o;
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const long long mr = mb + grow + 16 * it;
            const uint4 x = *(const uint4*)(scr + (grow + 16 * it) * 144 + gchunk * 16);
            if (mr < p.M) *(uint4*)(outp + mr * p.ldc + oc0 + gchunk * 8) = x;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      STAMP(ts4)
#ifdef LKGD_GEMM_STAMPS
      a_epi += ts4 - ts3;
#endif
    }
  }
#ifdef LKGD_GEMM_STAMPS
  if (t == 0 && blockIdx.x < 256) {
    unsigned long long t_end; STAMP(t_end)
    unsigned long long* o = lkgd_gemm_stamps + blockIdx.x * 8;
    o[0] = a_wait; o[1] = a_stage; o[2] = a_comp; o[3] = a_epi; o[4] = t_end - t_begin; o[5] = total;
  }
#endif
}

#ifdef LKGD_GEMM_STAMPS
extern "C" int lkgd_debug_read_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lkgd_gemm_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int lkgd_gemm_stream_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)lkgd_gemm_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SLDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  int tiles_m = (d->M + SBM - 1) / SBM, tiles_n = (d->N + SBN - 1) / SBN;
  long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  hipLaunchKernelGGL(lkgd_gemm_stream_kernel, dim3(grid), dim3(SNT), SLDS, stream, *d, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
