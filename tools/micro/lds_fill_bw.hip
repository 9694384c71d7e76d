// How fast can the chip fill LDS by LDS-DMA (global_load_lds_dwordx4) from an L2 / Infinity-Cache resident source - the
// operand-delivery roofline of the LDS-staged GEMM programs?  One persistent workgroup of 512 threads per CU streams 8 KiB
// chunks (one 16-byte load per thread) through a ring of `depth` chunks in flight: wait for the oldest (counted vmcnt), issue a new
// one.  No barrier, no compute, nothing reads the LDS: pure issue -> landed throughput.  Sweeps the bytes in flight per CU and the
// size of the source window (1 MiB per XCD slice ... 1 GiB: L2 -> MALL -> HBM).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_fill_bw.hip -o tools/micro/lds_fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, long long window, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, w = t >> 6;
  // every workgroup walks the window from its own start, 8 KiB per step, wrapping
  long long off = ((long long)blockIdx.x * 524288 + (long long)t * 16) % window;
  const long long step = 8192LL * 37;            // a stride that is not a multiple of the channel interleave
  auto issue = [&](int slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                     (__attribute__((address_space(3))) void*)(smem + slot * 8192 + w * 1024), 16, 0, 0);
    off += step;
    if (off >= window) off -= window;
  };
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    // the oldest of the DEPTH loads in flight has landed
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    issue(slot);
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0) clk[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
static void run(const char* src, long long window, int grid, unsigned long long* clk) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 8192);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(fill<DEPTH>, dim3(grid), dim3(512), DEPTH * 8192, 0, src, window, iters, clk);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)grid * (iters + DEPTH) * 8192.0;
  printf("  %3d KiB in flight per CU: %7.2f TB/s chip, %6.1f GB/s per CU, %5.2f us per 8 KiB chunk and CU\n", DEPTH * 8,
         bytes / best / 1e9, bytes / best / 1e6 / grid, best * 1e3 / (iters + DEPTH));
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount;
  char* src; unsigned long long* clk;
  const long long maxw = 1LL << 30;
  hipMalloc(&src, maxw + (1 << 20)); hipMemset(src, 1, maxw + (1 << 20)); hipMalloc(&clk, grid * 8);
  for (long long window : {8LL << 20, 128LL << 20, 1LL << 30}) {
    printf("source window %lld MiB (%s), %d workgroups of 512 threads:\n", window >> 20,
           window <= (32LL << 20) ? "fits the eight 4-MiB L2s" : window <= (256LL << 20) ? "fits the 256-MiB Infinity Cache" : "HBM", grid);
    run<1>(src, window, grid, clk); run<2>(src, window, grid, clk); run<4>(src, window, grid, clk);
    run<8>(src, window, grid, clk); run<16>(src, window, grid, clk);
  }
  return 0;
}
