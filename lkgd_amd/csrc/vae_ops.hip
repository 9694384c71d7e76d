// Clip-level VAE stages: the two small ops of AutoencoderKLTemporalDecoder [EXT diffusers 0.27.2] that
// are not GEMM / GroupNorm shaped (include/lkgd_hip.h section 12).
//   lkgd_softmax_rows      the single-head, head_dim-512 attention of the VAE mid blocks runs as GEMMs (QK^T, PV through
//                          lkgd_gemm_f16); this is the row softmax between them: one workgroup per score row, row in
//                          registers (16-byte loads), fp32 max / sum, wave shuffles + one LDS exchange.  HBM-bound:
//                          2 * S * S * 2 bytes per image.
//   lkgd_time_conv_out     `time_conv_out`: Conv3d (3,1,1) over the frames of a decode chunk on the 3 output channels,
//                          fused with the channels-last -> NCHW conversion of the decoded frames.  HBM-bound.
#include "common.h"

#define SM_NT 256

__global__ __launch_bounds__(SM_NT) void softmax_rows_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y,
                                                             int ldy, int cols) {
  __shared__ float red[SM_NT / 64];
  const long long row = blockIdx.x;
  const half_t* xr = x + row * ldx;
  half_t* yr = y + row * ldy;
  const int t = threadIdx.x;
  // each thread owns up to NV 8-element chunks, strided by the block (coalesced 16-byte accesses)
  constexpr int NV = 8;                     // cols <= 8 * 256 * 8 = 16384
  half8_t v[NV];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * SM_NT + t) * 8;
    if (c < cols) {
      v[i] = *(const half8_t*)(xr + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[i][e]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  const float mb = mx * 1.4426950408889634f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * SM_NT + t) * 8;
    if (c < cols) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf((float)v[i][e], 1.4426950408889634f, -mb));
        sum += p;
        v[i][e] = (half_t)p;            // p in [0, 1]; normalised below from the fp16-rounded value's fp32 twin
      }
    }
  }
  sum = wave_sum(sum);
  if ((t & 63) == 0) red[t >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * SM_NT + t) * 8;
    if (c < cols) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[i][e] * inv);
      *(half8_t*)(yr + c) = o;
    }
  }
}

extern "C" int lkgd_softmax_rows(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t cols,
                                 lkgd_stream_t stream) {
  if (!x || !y) return LKGD_E_NULL;
  if (rows <= 0 || cols <= 0 || cols > 8 * SM_NT * 8 || rows > 0x7fffffffLL) return LKGD_E_SHAPE;
  if (cols % 8 || ldx % 8 || ldy % 8 || ldx < cols || ldy < cols) return LKGD_E_ALIGN;
  if (!aligned16(x) || !aligned16(y)) return LKGD_E_ALIGN;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(SM_NT), 0, (hipStream_t)stream, (const half_t*)x, ldx,
                     (half_t*)y, ldy, cols);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// out[b, f, co, p] = bias[co] + sum_{kt, ci} w[co][ci][kt] * x[(b, f + kt - 1, p), ci]   (zero outside the chunk's frames)
__global__ __launch_bounds__(256) void time_conv_out_kernel(const half_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out32,
                                                            half_t* __restrict__ out16, int F, int HW, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (bf, p)
  if (i >= total) return;
  const long long bf = i / HW;
  const int p = (int)(i - bf * HW);
  const int f = (int)(bf % F);
  float acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int ff = f + kt - 1;
    if (ff < 0 || ff >= F) continue;
    const half_t* xp = x + (i + (long long)(kt - 1) * HW) * ldx;
    const half4_t xv = *(const half4_t*)xp;                                  // channels 0..2 (+1 pad), 8-byte aligned rows
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) acc[co] = fmaf(w[(co * 3 + ci) * 3 + kt], (float)xv[ci], acc[co]);
  }
#pragma unroll
  for (int co = 0; co < 3; ++co) {
    const long long o = (bf * 3 + co) * HW + p;
    if (out32) out32[o] = acc[co];
    else out16[o] = (half_t)acc[co];
  }
}

extern "C" int lkgd_time_conv_out(const void* tokens, int32_t ld, const float* w, const float* bias, void* out,
                                  int32_t out_is_f32, int64_t nbatch, int32_t F, int32_t HW, lkgd_stream_t stream) {
  if (!tokens || !w || !bias || !out) return LKGD_E_NULL;
  if (nbatch <= 0 || F <= 0 || HW <= 0 || ld < 4) return LKGD_E_SHAPE;
  if (ld % 4 || ((uintptr_t)tokens & 7)) return LKGD_E_ALIGN;
  const long long total = (long long)nbatch * F * HW;
  const long long nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(time_conv_out_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const half_t*)tokens, ld,
                     w, bias, out_is_f32 ? (float*)out : nullptr, out_is_f32 ? nullptr : (half_t*)out, F, HW, total);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
