// Practical MFMA ceiling on this box: register-only v_mfma_f32_32x32x16_f16 loop on every SIMD, timed with HIP events,
// plus the shader clock it ran at (s_memtime = shader cycles, s_memrealtime = 100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clk, int iters, int nacc) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
  float16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;
  int iters = 20000;
  int grid = 256 * wgs_per_cu;
  float* out; unsigned long long* clk;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, clk, iters, 4);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flop = (double)grid * 4 /*waves*/ * iters * 32.0 * 32768.0;
    printf("wg/cu %d rep %d: %.3f ms  %.1f TFLOP/s   shader cycles %llu / realtime ticks %llu -> %.0f MHz; cycles per MFMA %.2f\n",
           wgs_per_cu, rep, ms, flop / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 100.0, (double)h[0] / (iters * 32.0));
  }
  return 0;
}
