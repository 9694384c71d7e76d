// Issue cost of the softmax-side VALU instructions on one SIMD (one wave, independent instructions, s_memtime around the loop):
// v_exp_f32 vs v_exp_f16 vs v_cvt_pk_f16_f32 vs v_add_f32 vs v_max3_f32 vs v_dot2_f32_f16 vs v_pk_mul_f16.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(64) void rate(float* out, unsigned long long* clk, int iters) {
  float v0 = threadIdx.x * 0.01f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 * 0.5f, v5 = v0 * 0.25f, v6 = -v0, v7 = 0.125f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) asm volatile(REP16("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    if (OP == 1) asm volatile(REP16("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    if (OP == 2) asm volatile(REP16("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %5, %6\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_cvt_pk_f16_f32 %3, %7, %4\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if (OP == 3) asm volatile(REP16("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %7\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if (OP == 4) asm volatile(REP16("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max3_f32 %2, %2, %6, %7\n\tv_max3_f32 %3, %3, %7, %4\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if (OP == 5) asm volatile(REP16("v_dot2_f32_f16 %0, %4, %5, %0\n\tv_dot2_f32_f16 %1, %5, %6, %1\n\tv_dot2_f32_f16 %2, %6, %7, %2\n\tv_dot2_f32_f16 %3, %7, %4, %3\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if (OP == 6) asm volatile(REP16("v_pk_mul_f16 %0, %0, %4\n\tv_pk_mul_f16 %1, %1, %5\n\tv_pk_mul_f16 %2, %2, %6\n\tv_pk_mul_f16 %3, %3, %7\n\t") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    if (OP == 7) asm volatile(REP16("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t") : "+v"(*(double*)&v0), "+v"(*(double*)&v2) : "v"(*(double*)&v4), "v"(*(double*)&v6));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = v0 + v1 + v2 + v3;
  if (threadIdx.x == 0) clk[OP] = t1 - t0;
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256); hipMalloc(&clk, 64);
  const int iters = 20000;
  const char* names[] = {"v_exp_f32", "v_exp_f16", "v_cvt_pk_f16_f32", "v_add_f32", "v_max3_f32", "v_dot2_f32_f16", "v_pk_mul_f16"};
  hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<5>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(rate<6>, dim3(1), dim3(64), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 7; ++i) printf("%-18s %.2f cycles per instruction (one wave, 64 independent per iteration)\n", names[i], (double)h[i] / (iters * 64.0));
  return 0;
}
