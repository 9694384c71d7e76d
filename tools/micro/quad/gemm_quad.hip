// EXPERIMENT (round 3): does a FOUR-stage ring of BK = 32 half tiles remove the lock-step loss of the 256x320 program's K-loop
// (profiles/r03_wide_stamps.txt)?  Plain C = A . W^T, fp16 in / fp32 accumulate, 256x320 tile, 8 waves (wave tile 64x160), 160
// accumulators in a[0:159]; one workgroup per tile, no bias, direct stores.  Layout per stage (36 KiB): A image [256 rows][64 B] then
// W image [320 rows][64 B]; position (row r, 16-byte chunk c) holds source chunk c ^ (((r >> 3) & 1) << 1) (conflict-free
// ds_read_b128 of MFMA fragments; applied on the source side of the lane-linear LDS-DMA).
// Step s (half tile s):  wait own loads of half tile s+1 | barrier | issue half tile s+3 (inside the body, one per MFMA group) |
//                        40 MFMAs on stage s, and at their end the reads of half tile s+1's first fragments (stage s+1 landed
//                        before this step's barrier).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm tools/micro/quad/gemm_quad.hip -o tools/micro/quad/gemm_quad
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#ifdef QUAD_X_VLOAD           /* timing knob: the same bytes fetched into a (dead) register instead of LDS: is it the LDS write side? */
#define QUAD_LD(X, Y) Y
#else
#define QUAD_LD(X, Y) X
#endif
#include "gemm_quad_body.inc"

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
#define QST 36864            // bytes per stage
#define QA_BYTES 16384

template <int BASE>
__device__ __forceinline__ float4_t q_read_acc() {
  float a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\tv_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "i"(BASE));
  return (float4_t){a, b, c, d};
}
template <int V> struct QIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void q_for(F&& f, QIC<Is>...) { (f(QIC<Is>{}), ...); }

struct QIn {
  unsigned vA[2], vB[3];
  const char *sA, *sB;
  int m_a;
};

// MODE: 0 = NEXT, 1 = FIRST (C = 0);  LOAD: 2 / 3 = that many W groups for this wave, 0 = no loads this step
template <int FIRST, int LOAD>
__device__ __forceinline__ void q_body(half8_t (&xn)[4], half8_t& w0n, const half8_t (&xc)[4], half8_t& w0c, int wa, int xan, int wan,
                                       const QIn& in) {
  half8_t wt;
#define QSTMT(BODY)                                                                                                         \
  asm volatile(BODY                                                                                                         \
               : "=&v"(xn[0]), "=&v"(xn[1]), "=&v"(xn[2]), "=&v"(xn[3]), "=&v"(w0n), "=&v"(wt), "+v"(w0c)                   \
               : "v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]), "v"(wa), "v"(xan), "v"(wan), "v"(in.vA[0]), "v"(in.vA[1]), \
                 "v"(in.vB[0]), "v"(in.vB[1]), "v"(in.vB[2]), "s"(in.sA), "s"(in.sB), "s"(in.m_a)                           \
               : "memory", "scc", QUAD_AGPR_CLOBBERS)
  if (FIRST) {
    if (LOAD == 3) { QSTMT(QUAD_BODY_FIRST_NB3_EARLY); } else if (LOAD == 2) { QSTMT(QUAD_BODY_FIRST_NB2); }
    else if (LOAD == 12) { QSTMT(QUAD_BODY_FIRST_NB2_LATE); } else { QSTMT(QUAD_BODY_FIRST_NOLOAD); }
  } else {
    if (LOAD == 3) { QSTMT(QUAD_BODY_NEXT_NB3_EARLY); } else if (LOAD == 2) { QSTMT(QUAD_BODY_NEXT_NB2); }
    else if (LOAD == 12) { QSTMT(QUAD_BODY_NEXT_NB2_LATE); } else { QSTMT(QUAD_BODY_NEXT_NOLOAD); }
  }
#undef QSTMT
}

__device__ __forceinline__ void q_glds(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(96))) void gemm_quad_kernel(const half_t* __restrict__ A, int lda,
                                                                                               const half_t* __restrict__ W, int ldw,
                                                                                               half_t* __restrict__ C, int ldc, int M,
                                                                                               int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wr = w & 3, wc = w >> 2;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int nh = K / 32;
  const bool three = w < 4;                 // W groups w, w+8 and (w < 4) w+16 of the 20 groups of 16 rows
  // ---- LDS-DMA sources: lane l of a group -> row (l >> 2), position chunk (l & 3) holds source chunk (l & 3) ^ ((l >> 5) << 1)
  const int drow = lane >> 2, dch = (lane & 3) ^ ((lane >> 5) << 1);
  QIn in;
  {
    auto arow = [&](int g) { int r = tm * 256 + g * 16 + drow; return r < M ? r : M - 1; };
    in.vA[0] = (unsigned)arow(w) * (unsigned)lda * 2u + dch * 16;
    in.vA[1] = (unsigned)arow(w + 8) * (unsigned)lda * 2u + dch * 16;
#ifdef QUAD_X_WPACK          /* W pre-packed as LDS images [tile column][half tile][20 KiB]: every LDS-DMA instruction reads 1 KiB contiguous */
    in.vB[0] = (unsigned)(w * 1024 + lane * 16);
    in.vB[1] = (unsigned)((w + 8) * 1024 + lane * 16);
    in.vB[2] = (unsigned)(((three ? w + 16 : w)) * 1024 + lane * 16);
#else
    auto wrow = [&](int g) { return tn * 320 + g * 16 + drow; };
    in.vB[0] = (unsigned)wrow(w) * (unsigned)ldw * 2u + dch * 16;
    in.vB[1] = (unsigned)wrow(w + 8) * (unsigned)ldw * 2u + dch * 16;
    in.vB[2] = (unsigned)wrow(three ? w + 16 : w) * (unsigned)ldw * 2u + dch * 16;
#endif
  }
  // ---- fragment read addresses (stage 0): lane (l15, lq) reads row l15 of the fragment, position chunk lq ^ (((l15 >> 3) & 1) << 1)
  const int lane_sw = l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);
  const int xa0 = wr * 4096 + lane_sw, wa0 = QA_BYTES + wc * 10240 + lane_sw;
  auto stage_of = [](int s) { return (s & 3) * QST; };
  // ---- prologue: half tiles 0, 1, 2 by plain issue
  auto issue = [&](int s) {
    const char* a = (const char*)A + (size_t)s * 64;
#ifdef QUAD_X_WPACK
    const char* b = (const char*)W + ((size_t)tn * nh + s) * 20480;
#else
    const char* b = (const char*)W + (size_t)s * 64;
#endif
    char* d = smem + stage_of(s) + w * 1024;
    q_glds(a + in.vA[0], d);
    q_glds(a + in.vA[1], d + 8192);
    q_glds(b + in.vB[0], d + QA_BYTES);
    q_glds(b + in.vB[1], d + QA_BYTES + 8192);
    if (three) q_glds(b + in.vB[2], d + QA_BYTES + 16384);
  };
  for (int s = 0; s < 3 && s < nh; ++s) issue(s);
  // half tile 0 (and 1, when there is a third in flight behind them) landed
  if (nh >= 3) { if (three) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  half8_t xa[4], xb[4], wA, wB;
#pragma unroll
  for (int j = 0; j < 4; ++j) xa[j] = *(const half8_t*)(smem + xa0 + j * 1024);
  wA = *(const half8_t*)(smem + wa0);
  auto step = [&](int s, auto firstc, half8_t (&xc)[4], half8_t& w0c, half8_t (&xn)[4], half8_t& w0n) {
    constexpr int FIRST = decltype(firstc)::value;
#ifndef QUAD_X_NOBAR          /* timing knob: no wait, no barrier (results wrong) */
    if (s > 0) {
      // own loads of half tile s+1 landed: only half tile s+2's may still be in flight
#ifndef QUAD_X_NOWAIT         /* timing knob: the loads are issued but never waited for (results wrong) */
      if (s + 2 < nh) { if (three) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
    }
#endif
    const int cur = stage_of(s), nxt = stage_of(s + 1);
    in.sA = (const char*)A + (size_t)(s + 3) * 64;
#ifdef QUAD_X_WPACK
    in.sB = (const char*)W + ((size_t)tn * nh + (s + 3)) * 20480;
#else
    in.sB = (const char*)W + (size_t)(s + 3) * 64;
#endif
    in.m_a = stage_of(s + 3) + w * 1024;
#ifdef QUAD_X_NODMA           /* timing knob: the K-loop reads whatever the prologue left in LDS */
    if (false) {
#else
    if (s + 3 < nh) {
#endif
      if (three) q_body<FIRST, 3>(xn, w0n, xc, w0c, wa0 + cur, xa0 + nxt, wa0 + nxt, in);
#ifdef QUAD_X_LATE            /* experiment: waves 4-7 issue their loads in the second half of the body (waves 0-3 in the first) */
      else q_body<FIRST, 12>(xn, w0n, xc, w0c, wa0 + cur, xa0 + nxt, wa0 + nxt, in);
#else
      else q_body<FIRST, 2>(xn, w0n, xc, w0c, wa0 + cur, xa0 + nxt, wa0 + nxt, in);
#endif
    } else {
      q_body<FIRST, 0>(xn, w0n, xc, w0c, wa0 + cur, xa0 + nxt, wa0 + nxt, in);
    }
  };
  int s = 0;
  step(s, QIC<1>{}, xa, wA, xb, wB); ++s;
  for (; s + 1 < nh; s += 2) {
    step(s, QIC<0>{}, xb, wB, xa, wA);
    step(s + 1, QIC<0>{}, xa, wA, xb, wB);
  }
  if (s < nh) step(s, QIC<0>{}, xb, wB, xa, wA);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  // ---- epilogue: lane holds channels n0 + i*16 + 4*lq .. +3 of token row m0 + j*16 + l15
  const int n0 = tn * 320 + wc * 160 + 4 * lq;
  auto epi = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int m = tm * 256 + wr * 64 + j * 16 + l15;
    q_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const float4_t v = q_read_acc<(4 * i + j) * 4>();
      if (m < M) *(half4_t*)(C + (size_t)m * ldc + n0 + i * 16) = (half4_t){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    }, QIC<0>{}, QIC<1>{}, QIC<2>{}, QIC<3>{}, QIC<4>{}, QIC<5>{}, QIC<6>{}, QIC<7>{}, QIC<8>{}, QIC<9>{});
  };
  epi(QIC<0>{}); epi(QIC<1>{}); epi(QIC<2>{}); epi(QIC<3>{});
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
  hipFuncSetAttribute((const void*)gemm_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * QST);
  // ---- correctness on a small problem (one tile row x two tile columns, K = 192: six half tiles)
  {
    const int M = 256 + 40, N = 640, K = 192;
    std::vector<half_t> a((size_t)M * K), w((size_t)N * K), c((size_t)M * N);
    unsigned seed = 7;
    for (auto& v : a) v = (half_t)frand(seed);
    for (auto& v : w) v = (half_t)frand(seed);
    half_t *da, *dw, *dc;
    hipMalloc(&da, a.size() * 2); hipMalloc(&dw, w.size() * 2); hipMalloc(&dc, c.size() * 2);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
#ifdef QUAD_X_WPACK
    {
      std::vector<half_t> wp((size_t)N * K);
      const int nhh = K / 32;
      for (int tn_ = 0; tn_ < N / 320; ++tn_)
        for (int kh = 0; kh < nhh; ++kh)
          for (int r = 0; r < 320; ++r)
            for (int c = 0; c < 4; ++c) {
              const int sc = c ^ ((((r & 15) >> 3) & 1) << 1);          // position chunk c of row r holds source chunk sc
              for (int e = 0; e < 8; ++e)
                wp[((size_t)tn_ * nhh + kh) * 10240 + r * 32 + c * 8 + e] = w[(size_t)(tn_ * 320 + r) * K + kh * 32 + sc * 8 + e];
            }
      hipMemcpy(dw, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    }
#else
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
#endif
    hipMemset(dc, 0, c.size() * 2);
    const int tm = (M + 255) / 256, tn = N / 320;
    hipLaunchKernelGGL(gemm_quad_kernel, dim3(tm * tn), dim3(512), 4 * QST, 0, da, K, dw, K, dc, N, M, N, K, tn);
    hipDeviceSynchronize();
    hipMemcpy(c.data(), dc, c.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int m = 0; m < M; m += 7)
      for (int n = 0; n < N; n += 3) {
        float r = 0;
        for (int k = 0; k < K; ++k) r += (float)a[(size_t)m * K + k] * (float)w[(size_t)n * K + k];
        maxerr = fmax(maxerr, fabs(r - (float)c[(size_t)m * N + n]));
      }
    printf("check M=%d N=%d K=%d: max abs err %.4g (%s)\n", M, N, K, maxerr, maxerr < 2e-2 ? "ok" : "WRONG");
    hipFree(da); hipFree(dw); hipFree(dc);
#if !defined(QUAD_X_NODMA) && !defined(QUAD_X_NOBAR) && !defined(QUAD_X_NOWAIT) && !defined(QUAD_X_VLOAD)
    if (!(maxerr < 2e-2)) return 1;
#endif
  }
  // ---- timing: deep K and the model's linear shapes
#ifdef QUAD_X_RANDOM
  const int shapes[][3] = {{32768, 2560, 5120}, {16128, 1280, 5120}};
#else
  const int shapes[][3] = {{32768, 2560, 5120}, {64512, 1920, 640}, {64512, 640, 2560}, {16128, 1280, 5120}, {258048, 320, 1280}, {258048, 960, 320}};
#endif
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    half_t *da, *dw, *dc;
    hipMalloc(&da, (size_t)M * K * 2); hipMalloc(&dw, (size_t)N * K * 2); hipMalloc(&dc, (size_t)M * N * 2);
#ifdef QUAD_X_RANDOM          /* random operands (the chip's clock under MFMA load depends on the data: constant operands flatter) */
    {
      std::vector<half_t> ha((size_t)M * K), hw((size_t)N * K);
      unsigned sd = 99;
      for (auto& v : ha) v = (half_t)(frand(sd) * 2.0f);
      for (auto& v : hw) v = (half_t)(frand(sd) * 0.1f);
      hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    }
#else
    hipMemset(da, 0x11, (size_t)M * K * 2); hipMemset(dw, 0x22, (size_t)N * K * 2);
#endif
    const int tm = (M + 255) / 256, tn = N / 320;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it)
        hipLaunchKernelGGL(gemm_quad_kernel, dim3(tm * tn), dim3(512), 4 * QST, 0, da, K, dw, K, dc, N, M, N, K, tn);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = fminf(best, ms / 5);
    }
    printf("%7d x %5d x %5d: %.3f ms  %.0f TFLOP/s\n", M, N, K, best, 2.0 * M * N * K / best / 1e9);
    hipFree(da); hipFree(dw); hipFree(dc);
  }
  return 0;
}
