// Cross-attention of token rows against a SHORT per-context key / value set (include/lkgd_hip.h section 15): the literal
// form of BasicTransformerBlock.attn2 / TemporalBasicTransformerBlock.attn2 for a context of more than one token.  (SVD's
// context is ONE CLIP image token, for which the UNet folds attn2 into a row bias; this kernel is the path for Lk > 1.)
//
// Every query row attends to the Lk keys of the context its row map selects, independently of every other row, so the
// kernel is row-parallel: a workgroup = 256 query rows x one head; K and V of that head for ALL contexts
// (ncontexts * Lk rows x 64 channels, fp16) sit in LDS, read as broadcasts (the rows of a wave share one or two contexts);
// a thread keeps its query row (pre-scaled) and its output row in fp32 registers and runs an online softmax over the Lk
// keys.  Arithmetic intensity ~Lk/2 flop per byte of q / out and Lk is small (tens): HBM-bound, no MFMA.
#include "common.h"

#define XA_NT 256
#define XA_MAXKV 256          // ncontexts * Lk rows of K (and of V) per head in LDS: 2 x 32 KiB

__global__ __launch_bounds__(XA_NT) void attn_cross_kernel(const half_t* __restrict__ q, int ldq, const half_t* __restrict__ k,
                                                           int ldk, const half_t* __restrict__ v, int ldv,
                                                           half_t* __restrict__ out, int ldo, long long T, int nkv, int Lk,
                                                           int rb_d1, int rb_m1, int rb_d2, int rb_md, int rb_c0,
                                                           float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* sk = (half_t*)smem;                  // [nkv][64]
  half_t* sv = sk + (long long)nkv * 64;
  const int t = threadIdx.x, h = blockIdx.y;
  for (int c = t; c < nkv * 8; c += XA_NT) {   // 16-byte chunks of the head's K / V rows
    const int r = c >> 3, cc = c & 7;
    *(half8_t*)(sk + r * 64 + cc * 8) = *(const half8_t*)(k + (long long)r * ldk + h * 64 + cc * 8);
    *(half8_t*)(sv + r * 64 + cc * 8) = *(const half8_t*)(v + (long long)r * ldv + h * 64 + cc * 8);
  }
  __syncthreads();
  const long long m = (long long)blockIdx.x * XA_NT + t;
  if (m >= T) return;
  const int ctx = (int)(((m / rb_d1) * rb_m1 + m % rb_d2 + rb_c0) % rb_md);
  float qf[64], acc[64];
  {
    const half_t* qp = q + m * ldq + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const half8_t x = *(const half8_t*)(qp + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[i * 8 + e] = (float)x[e] * scale_log2e; acc[i * 8 + e] = 0.f; }
    }
  }
  float mx = -INFINITY, l = 0.f;
  const half_t* kp = sk + ctx * Lk * 64;
  const half_t* vp = sv + ctx * Lk * 64;
  for (int j = 0; j < Lk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const half8_t x = *(const half8_t*)(kp + j * 64 + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qf[i * 8 + e], (float)x[e], s);
    }
    const float mn = fmaxf(mx, s);
    const float corr = __builtin_amdgcn_exp2f(mx - mn), p = __builtin_amdgcn_exp2f(s - mn);
    mx = mn;
    l = l * corr + p;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const half8_t x = *(const half8_t*)(vp + j * 64 + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i * 8 + e] = fmaf(acc[i * 8 + e], corr, p * (float)x[e]);
    }
  }
  const float inv = 1.0f / l;
  half_t* op = out + m * ldo + h * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[i * 8 + e] * inv);
    *(half8_t*)(op + i * 8) = o;
  }
}

extern "C" int lkgd_attn_cross(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                               int32_t ldo, int64_t T, int32_t heads, int32_t ncontexts, int32_t Lk, int32_t rb_d1,
                               int32_t rb_m1, int32_t rb_d2, int32_t rb_md, int32_t rb_c0, float scale,
                               lkgd_stream_t stream) {
  if (!q || !k || !v || !out) return LKGD_E_NULL;
  if (T <= 0 || heads <= 0 || heads > 65535 || ncontexts <= 0 || Lk <= 0) return LKGD_E_SHAPE;
  if ((long long)ncontexts * Lk > XA_MAXKV) return LKGD_E_SHAPE;
  if (rb_d1 <= 0 || rb_d2 <= 0 || rb_md <= 0 || rb_m1 < 0 || rb_c0 < 0) return LKGD_E_SHAPE;
  {   // every row's context index must exist: bound of ((m / d1) * m1 + m % d2 + c0) % md over m < T
    const long long hi = ((T - 1) / rb_d1) * (long long)rb_m1 + ((T < rb_d2 ? T : rb_d2) - 1) + rb_c0;
    if ((hi < rb_md ? hi : (long long)rb_md - 1) >= ncontexts) return LKGD_E_SHAPE;
  }
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return LKGD_E_ALIGN;
  if (ldq < heads * 64 || ldk < heads * 64 || ldv < heads * 64 || ldo < heads * 64) return LKGD_E_SHAPE;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) return LKGD_E_ALIGN;
  const long long nblk = (T + XA_NT - 1) / XA_NT;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  const int nkv = ncontexts * Lk;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)attn_cross_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XA_MAXKV * 64 * 2 * 2) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  hipLaunchKernelGGL(attn_cross_kernel, dim3((unsigned)nblk, (unsigned)heads), dim3(XA_NT), (size_t)nkv * 64 * 2 * 2,
                     (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out,
                     ldo, (long long)T, nkv, Lk, rb_d1, rb_m1, rb_d2, rb_md, rb_c0, scale * 1.4426950408889634f);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
