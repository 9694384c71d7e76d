// Loop glue and small elementwise kernels (include/lkgd_hip.h sections 6, 7).  All HBM/latency-bound.
#include "common.h"

// ---- CFG duplicate + scale_model_input + channel concat, NCHW planes -> channels-last tokens [.., 8]
// reference: pipeline_stable_video_diffusion_trans.py:549-553, scheduler scale_model_input :284-285
template <typename LT>
__global__ __launch_bounds__(256) void prepare_input_kernel(const LT* __restrict__ latents,
                                                            const half_t* __restrict__ image_latents, int B, int F,
                                                            int HW, int cfg, float inv_scale,
                                                            half_t* __restrict__ out) {
  const long long total = (long long)cfg * B * F * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long nf = i / HW;          // cb*F + f
    const int f = (int)(nf % F);
    const int cb = (int)(nf / F);
    const int b = cb % B;                 // torch.cat([latents] * 2): uncond copies first, then cond copies
    half8_t o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = (float)latents[(((long long)b * F + f) * 4 + c) * HW + p];
      o[c] = (half_t)(x * inv_scale);
      o[4 + c] = image_latents[(((long long)cb * F + f) * 4 + c) * HW + p];
    }
    *(half8_t*)(out + i * 8) = o;
  }
}

// ---- per-frame CFG + Euler step (pipeline :578-592; scheduler.step :481-520).  fp16 rounding points follow the
// reference's tensor dtypes: CFG arithmetic on fp16 tensors, model_output * c_out stays fp16 (0-dim fp32 scalar does
// not promote), everything after the upcast of `sample` is fp32.
template <typename LT>
__global__ __launch_bounds__(256) void cfg_euler_kernel(const half_t* __restrict__ noise, LT* __restrict__ latents,
                                                        const float* __restrict__ guidance, int B, int F, int HW,
                                                        int cfg, float sigma, float sigma_next, int vpred) {
  const long long total = (long long)B * F * HW;
  const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
  const float c_skip = sigma * sigma + 1.0f;
  const float dt = sigma_next - sigma;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long bf = i / HW;
    const int f = (int)(bf % F);
    const int b = (int)(bf / F);
    half4_t u = *(const half4_t*)(noise + i * 4);
    half4_t n = u;
    if (cfg == 2) {
      half4_t c = *(const half4_t*)(noise + (i + total) * 4);
      const half_t g = (half_t)guidance[f];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t d = (half_t)((float)c[e] - (float)u[e]);
        half_t gd = (half_t)((float)g * (float)d);
        n[e] = (half_t)((float)u[e] + (float)gd);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long long li = (((long long)b * F + f) * 4 + c) * HW + p;
      const float x = (float)latents[li];
      float x0;
      if (vpred) x0 = (float)(half_t)((float)n[c] * c_out) + x / c_skip;
      else x0 = x - (float)(half_t)((float)n[c] * sigma);
      const float deriv = (x - x0) / sigma;
      latents[li] = (LT)(x + deriv * dt);
    }
  }
}

// ---- layout converters at the UNet.forward API boundary
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const half_t* __restrict__ tok, int ld, long long N,
                                                             int C, int HW, half_t* __restrict__ out) {
  const long long total = N * C * (long long)HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long nc = i / HW;
    const int c = (int)(nc % C);
    const long long n = nc / C;
    out[i] = tok[(n * HW + p) * ld + c];
  }
}
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const half_t* __restrict__ in, long long N, int C, int HW,
                                                             half_t* __restrict__ tok, int ld) {
  const long long total = N * C * (long long)HW;
  // consecutive threads walk p (coalesced reads from the planes); writes are strided by ld (small C at this boundary)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long nc = i / HW;
    const int c = (int)(nc % C);
    const long long n = nc / C;
    tok[(n * HW + p) * ld + c] = in[i];
  }
}

// ---- sinusoidal timestep embedding [cos | sin] (flip_sin_to_cos=True, shift 0): unet_..._controlnet.py:406,415
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, half_t* __restrict__ out,
                                          int ldo) {
  const int half_dim = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * half_dim; i += gridDim.x * blockDim.x) {
    const int r = i / half_dim, j = i - r * half_dim;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half_dim);
    const float a = t[r] * freq;
    out[(long long)r * ldo + j] = (half_t)cosf(a);
    out[(long long)r * ldo + half_dim + j] = (half_t)sinf(a);
  }
}

__global__ __launch_bounds__(256) void silu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long long n) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * 2048) {
    if (i + 8 <= n) {
      half8_t v = *(const half8_t*)(x + i), o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)silu_f((float)v[e]);
      *(half8_t*)(y + i) = o;
    } else {
      for (long long j = i; j < n; ++j) y[j] = (half_t)silu_f((float)x[j]);
    }
  }
}
__global__ __launch_bounds__(256) void add_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                  half_t* __restrict__ y, long long n) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * 2048) {
    if (i + 8 <= n) {
      *(half8_t*)(y + i) = *(const half8_t*)(a + i) + *(const half8_t*)(b + i);
    } else {
      for (long long j = i; j < n; ++j) y[j] = a[j] + b[j];
    }
  }
}

// ---- standalone scheduler API kernels (scale_model_input :264-288 / step :418-528 on tensors of any shape)
__global__ __launch_bounds__(256) void scale_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long long n,
                                                    float s) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = (half_t)((float)x[i] * s);
}
template <typename ST>
__global__ __launch_bounds__(256) void euler_kernel(const half_t* __restrict__ mo, const ST* __restrict__ sample,
                                                    const half_t* __restrict__ noise, half_t* __restrict__ prev, long long n,
                                                    float sigma, float sigma_hat, float s_noise, float churn, float sigma_next,
                                                    int vpred) {
  // sigma_hat = sigma * (gamma + 1); gamma > 0 ("churn", scheduling_euler_discrete_karras_fix.py:485-497) first adds
  // noise * s_noise * sqrt(sigma_hat^2 - sigma^2) to the sample, each product rounded to fp16 as the reference's tensors are
  const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
  const float c_skip = sigma * sigma + 1.0f;
  const float dt = sigma_next - sigma_hat;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float x = (float)sample[i];
    if (noise) x += (float)(half_t)((float)(half_t)((float)noise[i] * s_noise) * churn);
    float x0;
    if (vpred) x0 = (float)(half_t)((float)mo[i] * c_out) + x / c_skip;
    else x0 = x - (float)(half_t)((float)mo[i] * sigma_hat);
    prev[i] = (half_t)(x + (x - x0) / sigma_hat * dt);
  }
}

static unsigned grid_for(long long work_items, int per_block) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" int lkgd_prepare_unet_input(const void* latents, int32_t latents_is_f32, const void* image_latents,
                                       int32_t B, int32_t F, int32_t H, int32_t W, int32_t cfg, float sigma,
                                       void* tokens_out, lkgd_stream_t stream) {
  if (!latents || !image_latents || !tokens_out) return LKGD_E_NULL;
  if (B <= 0 || F <= 0 || H <= 0 || W <= 0 || (cfg != 1 && cfg != 2)) return LKGD_E_SHAPE;
  if (!aligned16(tokens_out)) return LKGD_E_ALIGN;
  const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
  const long long total = (long long)cfg * B * F * H * W;
  if (latents_is_f32)
    hipLaunchKernelGGL(prepare_input_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)latents, (const half_t*)image_latents, B, F, H * W, cfg, inv,
                       (half_t*)tokens_out);
  else
    hipLaunchKernelGGL(prepare_input_kernel<half_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)latents, (const half_t*)image_latents, B, F, H * W, cfg, inv,
                       (half_t*)tokens_out);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_cfg_euler_step(const void* noise_tokens, void* latents, int32_t latents_is_f32,
                                   const float* guidance, int32_t B, int32_t F, int32_t H, int32_t W, int32_t cfg,
                                   float sigma, float sigma_next, int32_t prediction_type, lkgd_stream_t stream) {
  if (!noise_tokens || !latents) return LKGD_E_NULL;
  if (cfg == 2 && !guidance) return LKGD_E_NULL;
  if (B <= 0 || F <= 0 || H <= 0 || W <= 0 || (cfg != 1 && cfg != 2) || !(sigma > 0.f)) return LKGD_E_SHAPE;
  if (prediction_type != 0 && prediction_type != 1) return LKGD_E_MODE;
  if ((uintptr_t)noise_tokens & 7) return LKGD_E_ALIGN;
  const long long total = (long long)B * F * H * W;
  if (latents_is_f32)
    hipLaunchKernelGGL(cfg_euler_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)noise_tokens, (float*)latents, guidance, B, F, H * W, cfg, sigma, sigma_next,
                       prediction_type);
  else
    hipLaunchKernelGGL(cfg_euler_kernel<half_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)noise_tokens, (half_t*)latents, guidance, B, F, H * W, cfg, sigma, sigma_next,
                       prediction_type);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// ---- rows of a frame shard regrouped by destination pixel shard (and back): the pack / unpack around the all-to-all that
// re-shards the temporal attention by pixels (lkgd_amd/dist.py::frames_to_pixels / pixels_to_frames).  Row (f, p) of the
// [fl, HW, C] slice sits in the send buffer at  fl * p0[r] + f * px[r] + (p - p0[r])  with r = the shard that owns pixel p.
typedef unsigned sr_vec_t __attribute__((ext_vector_type(4)));
struct shard_tab { int k; int px[16]; int p0[16]; };
template <bool PACK>
__global__ __launch_bounds__(256) void shard_rows_kernel(const sr_vec_t* __restrict__ src, sr_vec_t* __restrict__ dst, int fl, int HW,
                                                         int c8, shard_tab tab) {
  const long long n = (long long)fl * HW * c8;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long row = i / c8;
    const int v = (int)(i - row * c8);
    const int f = (int)(row / HW), p = (int)(row - (long long)f * HW);
    int r = 0;
    while (r + 1 < tab.k && p >= tab.p0[r + 1]) ++r;
    const long long prow = (long long)fl * tab.p0[r] + (long long)f * tab.px[r] + (p - tab.p0[r]);
    if (PACK) dst[prow * c8 + v] = src[i];
    else dst[i] = src[prow * c8 + v];
  }
}

extern "C" int lkgd_shard_rows(const void* src, void* dst, int32_t fl, int32_t HW, int32_t C, int32_t k, const int32_t* px,
                               int32_t pack, lkgd_stream_t stream) {
  if (!src || !dst || !px) return LKGD_E_NULL;
  if (fl <= 0 || HW <= 0 || C <= 0 || C % 8 || k <= 0 || k > 16) return LKGD_E_SHAPE;
  if (!aligned16(src) || !aligned16(dst)) return LKGD_E_ALIGN;
  shard_tab tab;
  tab.k = k;
  int o = 0;
  for (int r = 0; r < k; ++r) {
    if (px[r] <= 0) return LKGD_E_SHAPE;
    tab.px[r] = px[r];
    tab.p0[r] = o;
    o += px[r];
  }
  if (o != HW) return LKGD_E_SHAPE;
  const long long n = (long long)fl * HW * (C / 8);
  const dim3 gr(grid_for(n, 256)), bl(256);
  if (pack) hipLaunchKernelGGL(shard_rows_kernel<true>, gr, bl, 0, (hipStream_t)stream, (const sr_vec_t*)src, (sr_vec_t*)dst, fl, HW, C / 8, tab);
  else hipLaunchKernelGGL(shard_rows_kernel<false>, gr, bl, 0, (hipStream_t)stream, (const sr_vec_t*)src, (sr_vec_t*)dst, fl, HW, C / 8, tab);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_tokens_to_nchw(const void* tokens, int32_t ld, int64_t N, int32_t C, int32_t HW, void* out,
                                   lkgd_stream_t stream) {
  if (!tokens || !out) return LKGD_E_NULL;
  if (N <= 0 || C <= 0 || HW <= 0 || ld < C) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3(grid_for(N * C * (long long)HW, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)tokens, ld, (long long)N, C, HW, (half_t*)out);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_nchw_to_tokens(const void* nchw, int64_t N, int32_t C, int32_t HW, void* tokens, int32_t ld,
                                   lkgd_stream_t stream) {
  if (!nchw || !tokens) return LKGD_E_NULL;
  if (N <= 0 || C <= 0 || HW <= 0 || ld < C) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3(grid_for(N * C * (long long)HW, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)nchw, (long long)N, C, HW, (half_t*)tokens, ld);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_timestep_embedding(const float* t, int32_t n, int32_t dim, void* out, int32_t ldo,
                                       lkgd_stream_t stream) {
  if (!t || !out) return LKGD_E_NULL;
  if (n <= 0 || dim <= 0 || dim % 2 || ldo < dim) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long long)n * dim / 2, 256)), dim3(256), 0,
                     (hipStream_t)stream, t, n, dim, (half_t*)out, ldo);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_silu(const void* x, void* y, int64_t n, lkgd_stream_t stream) {
  if (!x || !y) return LKGD_E_NULL;
  if (n <= 0) return LKGD_E_SHAPE;
  if (!aligned16(x) || !aligned16(y)) return LKGD_E_ALIGN;
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (half_t*)y, (long long)n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_add(const void* a, const void* b, void* y, int64_t n, lkgd_stream_t stream) {
  if (!a || !b || !y) return LKGD_E_NULL;
  if (n <= 0) return LKGD_E_SHAPE;
  if (!aligned16(a) || !aligned16(b) || !aligned16(y)) return LKGD_E_ALIGN;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, (const half_t*)a,
                     (const half_t*)b, (half_t*)y, (long long)n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_scale(const void* x, void* y, int64_t n, float s, lkgd_stream_t stream) {
  if (!x || !y) return LKGD_E_NULL;
  if (n <= 0) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                     (half_t*)y, (long long)n, s);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

static int euler_launch(const void* model_output, const void* sample, int32_t sample_is_f32, const void* noise, void* prev,
                        int64_t n, float sigma, float sigma_hat, float s_noise, float churn, float sigma_next,
                        int32_t prediction_type, lkgd_stream_t stream) {
  if (!model_output || !sample || !prev) return LKGD_E_NULL;
  if (n <= 0 || !(sigma > 0.f) || !(sigma_hat >= sigma)) return LKGD_E_SHAPE;
  if (prediction_type != 0 && prediction_type != 1) return LKGD_E_MODE;
  if (sample_is_f32)
    hipLaunchKernelGGL(euler_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)model_output, (const float*)sample, (const half_t*)noise, (half_t*)prev, (long long)n,
                       sigma, sigma_hat, s_noise, churn, sigma_next, prediction_type);
  else
    hipLaunchKernelGGL(euler_kernel<half_t>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)model_output, (const half_t*)sample, (const half_t*)noise, (half_t*)prev, (long long)n,
                       sigma, sigma_hat, s_noise, churn, sigma_next, prediction_type);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_euler_step(const void* model_output, const void* sample, int32_t sample_is_f32, void* prev,
                               int64_t n, float sigma, float sigma_next, int32_t prediction_type,
                               lkgd_stream_t stream) {
  return euler_launch(model_output, sample, sample_is_f32, nullptr, prev, n, sigma, sigma, 0.f, 0.f, sigma_next,
                      prediction_type, stream);
}

extern "C" int lkgd_euler_step_churn(const void* model_output, const void* sample, int32_t sample_is_f32, const void* noise,
                                     void* prev, int64_t n, float sigma, float sigma_hat, float s_noise, float churn,
                                     float sigma_next, int32_t prediction_type, lkgd_stream_t stream) {
  if (!noise) return LKGD_E_NULL;
  return euler_launch(model_output, sample, sample_is_f32, noise, prev, n, sigma, sigma_hat, s_noise, churn, sigma_next,
                      prediction_type, stream);
}

extern "C" const char* lkgd_version(void) { return "lkgd_hip 1 gfx950"; }

// ---------------------------------------------------------------------------------------------------------------
// DiT glue (include/lkgd_hip.h section 14; CogVideoX blocks)
//   lkgd_gelu_tanh : y = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))   F.gelu(approximate="tanh"), fp16 in / out, in place ok
//   lkgd_gated_add : out[r, :] = res[r, :] + gate[g(r), :] * x[r, :],  g(r) = (r / rows_per_batch) * 2 + ((r % rows_per_batch) >= split)
//                    the adaLN-zero residual of the two token streams (text rows first, then video rows) of every batch entry
__global__ void gelu_tanh_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const half8_t v = *(const half8_t*)(x + i * 8);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)v[e];
    const float u = 0.7978845608028654f * (f + 0.044715f * f * f * f);
    // tanh(u) = 1 - 2 / (exp(2u) + 1); exp2-based, saturates cleanly for |u| large
    const float t = 1.0f - 2.0f / (__builtin_amdgcn_exp2f(u * 2.885390081777927f) + 1.0f);
    o[e] = (half_t)(0.5f * f * (1.0f + t));
  }
  *(half8_t*)(y + i * 8) = o;
}

extern "C" int lkgd_gelu_tanh(const void* x, void* y, int64_t n, lkgd_stream_t stream) {
  if (!x || !y) return LKGD_E_NULL;
  if (n <= 0 || n % 8) return LKGD_E_SHAPE;
  if (!aligned16(x) || !aligned16(y)) return LKGD_E_ALIGN;
  const long long n8 = n / 8, nblk = (n8 + 255) / 256;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(gelu_tanh_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, n8);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

__global__ void gated_add_kernel(const half_t* __restrict__ x, int ldx, const float* __restrict__ gate, const half_t* __restrict__ res,
                                 int ldr, half_t* __restrict__ out, int ldo, long long rows, int C8, int rows_per_batch, int split) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C8) return;
  const long long r = i / C8;
  const int cv = (int)(i - r * C8);
  const long long b = r / rows_per_batch;
  const int g = (int)(b * 2 + ((r - b * rows_per_batch) >= split ? 1 : 0));
  const half8_t xv = *(const half8_t*)(x + r * ldx + cv * 8), rv = *(const half8_t*)(res + r * ldr + cv * 8);
  const float4_t g0 = *(const float4_t*)(gate + ((long long)g * C8 + cv) * 8), g1 = *(const float4_t*)(gate + ((long long)g * C8 + cv) * 8 + 4);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)rv[e] + (e < 4 ? g0[e & 3] : g1[e & 3]) * (float)xv[e]);
  *(half8_t*)(out + r * ldo + cv * 8) = o;
}

extern "C" int lkgd_gated_add(const void* x, int32_t ldx, const float* gate, const void* res, int32_t ldr, void* out, int32_t ldo,
                              int64_t rows, int32_t C, int32_t rows_per_batch, int32_t split, lkgd_stream_t stream) {
  if (!x || !gate || !res || !out) return LKGD_E_NULL;
  if (rows <= 0 || C <= 0 || C % 8 || rows_per_batch <= 0 || rows % rows_per_batch || split < 0 || split > rows_per_batch)
    return LKGD_E_SHAPE;
  if (ldx % 8 || ldr % 8 || ldo % 8 || !aligned16(x) || !aligned16(res) || !aligned16(out) || !aligned16(gate)) return LKGD_E_ALIGN;
  const long long total = rows * (C / 8), nblk = (total + 255) / 256;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(gated_add_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, gate,
                     (const half_t*)res, ldr, (half_t*)out, ldo, (long long)rows, C / 8, rows_per_batch, split);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
