// Store throughput of the chip as a function of how many CUs store at once: G persistent workgroups of 512 threads each
// write BYTES_PER_WG bytes of their own region, as 8-byte pieces in the accumulator layout of the 256x320 GEMM epilogue
// (16 rows x 32 bytes per wave instruction) or as 16-byte pieces of contiguous rows.  Is the ~7 TB/s a fill reaches a chip
// limit (then G = 64 gives 4x the per-CU rate) or a per-CU limit (then the per-CU rate does not depend on G)?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_bw.hip -o gpurun_out/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void store_loop(_Float16* out, long long rows_per_wg, int ldc, int xcd_mask = 0xff) {
  if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;         // only the workgroups of some XCDs store (round-robin dispatch)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  _Float16* base = out + (long long)blockIdx.x * rows_per_wg * ldc;
  half8 v8; half4 v4;
  for (int i = 0; i < 8; ++i) v8[i] = (_Float16)(t + i);
  for (int i = 0; i < 4; ++i) v4[i] = (_Float16)(t + i);
  if (MODE == 0) {            // wave w: rows [r0 + w*16, +16), lane (l15, lq) writes 4 channels at column 4*lq + 16*i
    const int l15 = lane & 15, lq = lane >> 4;
    for (long long r0 = 0; r0 < rows_per_wg; r0 += 128) {
      _Float16* p = base + (r0 + w * 16 + l15) * ldc + 4 * lq;
#pragma unroll
      for (int i = 0; i < 20; ++i) *(half4*)(p + i * 16) = v4;       // 320 columns
    }
  } else if (MODE == 2) {     // accumulator layout after a lane-pair exchange: 16 bytes per lane, 64 contiguous bytes per row and instruction
    const int l15 = lane & 15, lq = lane >> 4;
    for (long long r0 = 0; r0 < rows_per_wg; r0 += 128) {
      _Float16* p = base + (r0 + w * 16 + l15) * ldc + (lq & 1) * 16 + (lq >> 1) * 8;
#pragma unroll
      for (int i = 0; i < 10; ++i) *(half8*)(p + i * 32) = v8;       // 320 columns
    }
  } else {                    // wave w: rows [r0 + w*16, +16) as 16-byte pieces, 40 pieces per 320-column row
    for (long long r0 = 0; r0 < rows_per_wg; r0 += 128) {
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int c = lane + 64 * k, row = c / 40, cc = c - row * 40;
        *(half8*)(base + (r0 + w * 16 + row) * ldc + cc * 8) = v8;
      }
    }
  }
}

int main() {
  const int ldc = 320;
  const long long total_rows = 256LL * 4096;      // 640 MiB
  _Float16* out; hipMalloc(&out, total_rows * ldc * 2);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int mode = 0; mode < 3; ++mode)
    for (int G : {16, 32, 64, 128, 256, 512, 1024}) {
      long long rows = total_rows / 1024;          // every grid writes the same bytes per workgroup
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s);
        if (mode == 0) hipLaunchKernelGGL(store_loop<0>, dim3(G), dim3(512), 0, 0, out, rows, ldc);
        else if (mode == 1) hipLaunchKernelGGL(store_loop<1>, dim3(G), dim3(512), 0, 0, out, rows, ldc);
        else hipLaunchKernelGGL(store_loop<2>, dim3(G), dim3(512), 0, 0, out, rows, ldc);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
      }
      double bytes = (double)G * rows * ldc * 2;
      printf("%s  G %4d: %7.3f ms  %6.2f TB/s  = %6.1f GB/s per workgroup\n", mode == 1 ? "16-byte row pieces " : mode == 2 ? "16-byte paired lanes" : "8-byte acc layout   ", G, best,
             bytes / best / 1e9, bytes / best / 1e6 / G);
    }
  // all CUs of 4 or 2 XCDs storing (16-byte row pieces): is the write limit the chip's or each XCD's?
  for (int mask : {0xff, 0x0f, 0x55, 0x03, 0x01}) {
    long long rows = total_rows / 1024;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(s);
      hipLaunchKernelGGL(store_loop<1>, dim3(256), dim3(512), 0, 0, out, rows, ldc, mask);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      if (ms < best) best = ms;
    }
    int n = __builtin_popcount(mask) * 32;
    double bytes = (double)n * rows * ldc * 2;
    printf("XCD mask 0x%02x (%3d workgroups): %7.3f ms  %6.2f TB/s  = %6.1f GB/s per workgroup\n", mask, n, best, bytes / best / 1e9,
           bytes / best / 1e6 / n);
  }
  return 0;
}
