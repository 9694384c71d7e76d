// Register-fed MFMA streams on random / constant fp16 operands: what the matrix pipe sustains (clock under load included) with
// v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 at the same FLOPs per wave (160 accumulator registers, two waves per
// SIMD, one 512-thread workgroup per CU).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32 (40 accumulators of 4), 1: 32x32x16 (10 accumulators of 16)
__global__ __launch_bounds__(512, 2) void mfma_loop(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 512 + threadIdx.x;
  half8 a[4], b[10];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = src[(t * 14 + i) & 0xfffff];
#pragma unroll
  for (int i = 0; i < 10; ++i) b[i] = src[(t * 14 + 4 + i) & 0xfffff];
  float acc_sum = 0.f;
  if (SHAPE == 0) {
    float4v c[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) c[i] = (float4v){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i], a[j], c[i * 4 + j], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) acc_sum += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else {
    float16v c[10];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // the same FLOPs per iteration: 10 accumulators x 2 k-steps of 16 = 20 instructions of 32768 FLOP
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 10; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i], a[(i + ks) & 3], c[i], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc_sum += c[i][e];
  }
  out[t] = acc_sum;
}

int main() {
  const int n = 1 << 20;
  std::vector<_Float16> h((size_t)n * 8);
  half8* src; float* out;
  hipMalloc(&src, (size_t)n * 16); hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) h[i] = kind == 0 ? (_Float16)0.5f : (_Float16)((rand() % 2001 - 1000) * 0.001f);
    hipMemcpy(src, h.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      const int iters = 20000;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(256), dim3(512), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(256), dim3(512), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double flop = 256.0 * 8 * iters * 40 * 16384.0;
      printf("%s operands, %s: %.3f ms  %.0f TFLOP/s\n", kind == 0 ? "constant" : "random  ", shape == 0 ? "16x16x32" : "32x32x16", best,
             flop / best / 1e9);
    }
  }
  return 0;
}
