// Clip-level image pre-processing of the CLIP branch (include/lkgd_hip.h section 11): the anti-aliased resize the
// reference applies to the conditioning frame before the CLIP feature extractor
// (pipeline/pipeline_stable_video_diffusion_trans.py:661-765: separable Gaussian blur with reflect padding, then bicubic
// interpolation with align_corners=True).  fp32 NCHW planes, once per clip, a few MB: plain one-thread-per-output kernels,
// bound by launch latency; nothing here is on the per-step path.
#include "common.h"

__device__ __forceinline__ int reflect_index(int i, int n) {          // F.pad(mode="reflect"): -1 -> 1, n -> n-2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <int AXIS>   // 1: along W, 0: along H
__global__ __launch_bounds__(256) void conv1d_reflect_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long long total, int H, int W,
                                                            const float* __restrict__ taps, int ntaps) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long long r = i / W;
  const int y = (int)(r % H);
  const long long plane = r / H;
  const int front = (ntaps - 1) / 2;
  const float* base = in + plane * H * W;
  float acc = 0.f;
  for (int j = 0; j < ntaps; ++j) {
    float v;
    if (AXIS == 1) v = base[(long long)y * W + reflect_index(x + j - front, W)];
    else v = base[(long long)reflect_index(y + j - front, H) * W + x];
    acc = fmaf(v, taps[j], acc);
  }
  out[i] = acc;
}

// cubic convolution coefficients, A = -0.75 (torch upsample_bicubic2d)
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void resize_bicubic_ac_kernel(const float* __restrict__ in, long long planes, int H,
                                                               int W, float* __restrict__ out, int Ho, int Wo,
                                                               float sy, float sx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= planes * Ho * Wo) return;
  const int ox = (int)(i % Wo);
  const long long r = i / Wo;
  const int oy = (int)(r % Ho);
  const long long plane = r / Ho;
  const float fy = sy * (float)oy, fx = sx * (float)ox;      // align_corners=True: src = dst * (in-1)/(out-1)
  const int iy = (int)floorf(fy), ix = (int)floorf(fx);
  float cy[4], cx[4];
  cubic_coeffs(fy - (float)iy, cy);
  cubic_coeffs(fx - (float)ix, cx);
  const float* base = in + plane * H * W;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int yy = iy - 1 + a;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    const float* row = base + (long long)yy * W;
    float rowv = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int xx = ix - 1 + b;
      xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
      rowv += row[xx] * cx[b];
    }
    acc += rowv * cy[a];
  }
  out[i] = acc;
}

extern "C" int lkgd_conv1d_reflect(const float* in, float* out, int64_t planes, int32_t H, int32_t W, const float* taps,
                                   int32_t ntaps, int32_t axis, lkgd_stream_t stream) {
  if (!in || !out || !taps) return LKGD_E_NULL;
  if (planes <= 0 || H <= 0 || W <= 0 || ntaps <= 0 || in == out) return LKGD_E_SHAPE;
  if (axis != 0 && axis != 1) return LKGD_E_MODE;
  if (ntaps / 2 >= (axis ? W : H)) return LKGD_E_SHAPE;                 // reflect padding needs pad < size
  const long long total = (long long)planes * H * W;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (axis == 1)
    hipLaunchKernelGGL(conv1d_reflect_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, total, H, W,
                       taps, ntaps);
  else
    hipLaunchKernelGGL(conv1d_reflect_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, total, H, W,
                       taps, ntaps);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_resize_bicubic_ac(const float* in, int64_t planes, int32_t H, int32_t W, float* out, int32_t Ho,
                                      int32_t Wo, lkgd_stream_t stream) {
  if (!in || !out) return LKGD_E_NULL;
  if (planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return LKGD_E_SHAPE;
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)planes * Ho * Wo;
  hipLaunchKernelGGL(resize_bicubic_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, (long long)planes, H, W, out, Ho, Wo, sy, sx);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------
// ViT front end of the domain / flow encoders (include/lkgd_hip.h section 13): F.interpolate(mode="bilinear",
// align_corners=False) to S x S fused with the P x P patch unfold of the patch-embedding convolution - the resized image
// never exists; out[(n, py, px), c*P*P + ky*P + kx] fp16 is the A operand of the patch-embedding GEMM.
__global__ void vit_patchify_kernel(const float* __restrict__ in, int C, int H, int W, half_t* __restrict__ out, int S, int P,
                                    float sy, float sx, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int K = C * P * P, g = S / P;
  const long long row = i / K;
  const int k = (int)(i - row * K);
  const int c = k / (P * P), kk = k - c * P * P, ky = kk / P, kx = kk - ky * P;
  const long long n = row / (g * g);
  const int pr = (int)(row - n * g * g), py = pr / g, px = pr - py * g;
  const int y = py * P + ky, x = px * P + kx;
  // PyTorch's area_pixel_compute_source_index(scale, dst, align_corners=False, cubic=False): max(scale*(dst+0.5)-0.5, 0)
  float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 > H - 1 ? H - 1 : y0;
  x0 = x0 > W - 1 ? W - 1 : x0;
  const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* b = in + (n * C + c) * (long long)H * W;
  const float v = (1.f - ly) * ((1.f - lx) * b[(long long)y0 * W + x0] + lx * b[(long long)y0 * W + x1]) +
                  ly * ((1.f - lx) * b[(long long)y1 * W + x0] + lx * b[(long long)y1 * W + x1]);
  out[i] = (half_t)v;
}

extern "C" int lkgd_vit_patchify(const float* in, int64_t nimg, int32_t C, int32_t H, int32_t W, void* out, int32_t S,
                                 int32_t P, lkgd_stream_t stream) {
  if (!in || !out) return LKGD_E_NULL;
  if (nimg <= 0 || C <= 0 || H <= 0 || W <= 0 || S <= 0 || P <= 0 || S % P) return LKGD_E_SHAPE;
  const long long total = (long long)nimg * S * S * C;
  const long long nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(vit_patchify_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, in, C, H, W, (half_t*)out, S,
                     P, (float)H / (float)S, (float)W / (float)S, total);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
