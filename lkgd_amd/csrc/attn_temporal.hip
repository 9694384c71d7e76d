// Temporal self-attention over F <= 32 frames, head_dim 64 (include/lkgd_hip.h section 5).
//
// HBM-bound (14 keys per query: arithmetic intensity ~7 flop/B), so no MFMA: one thread owns one
// (batch, pixel, head, query-frame) row.  A 256-thread workgroup = 16 (pixel, head) pairs x 16 query-frame slots,
// pairs on the fast lane index, so that
//   * every global access is a run of consecutive 128-byte head segments (16 lanes x 128 B = 2 KiB contiguous), and
//   * the F query threads that share one pair's K/V rows sit in the same workgroup: K/V reach HBM once and are
//     re-served from the CU's L1.
// The [B*F,S,C] <-> [B*S,F,C] regroup of the reference is this kernel's index map; nothing is copied.
#include "common.h"

template <int FMAX>
__global__ __launch_bounds__(FMAX * 16) void attn_temporal_kernel(const half_t* __restrict__ q, int ldq,
                                                            const half_t* __restrict__ k, int ldk,
                                                            const half_t* __restrict__ v, int ldv,
                                                            half_t* __restrict__ out, int ldo, int Fq, int F, int S,
                                                            int heads, const int* __restrict__ kvmap,
                                                            float scale_log2e) {
  const int t = threadIdx.x;
  const int fq = t >> 4;
  const long long pid = (long long)blockIdx.x * 16 + (t & 15);
  const int b = blockIdx.y;
  const long long npairs = (long long)S * heads;
  if (pid >= npairs || fq >= Fq) return;
  const int s = (int)(pid / heads), hh = (int)(pid - (long long)s * heads);
  const int kvb = kvmap ? kvmap[b] : b;

  const half_t* qp = q + (((long long)b * Fq + fq) * S + s) * ldq + hh * 64;
  half8_t qv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qv[i] = *(const half8_t*)(qp + i * 8);

  float sc[FMAX];
  float mx = -1e30f;
#pragma unroll
  for (int f = 0; f < FMAX; ++f) {
    sc[f] = -1e30f;
    if (f < F) {
      const half_t* kp = k + (((long long)kvb * F + f) * S + s) * ldk + hh * 64;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        half8_t kv = *(const half8_t*)(kp + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          half2_t a = {qv[i][2 * e], qv[i][2 * e + 1]};
          half2_t bb = {kv[2 * e], kv[2 * e + 1]};
          acc = __builtin_amdgcn_fdot2(a, bb, acc, false);
        }
      }
      sc[f] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  float l = 0.f;
  const float mb = mx * scale_log2e;
#pragma unroll
  for (int f = 0; f < FMAX; ++f) {
    float p = f < F ? __builtin_amdgcn_exp2f(fmaf(sc[f], scale_log2e, -mb)) : 0.f;
    sc[f] = p;
    l += p;
  }
  const float inv = 1.0f / l;
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
#pragma unroll
  for (int f = 0; f < FMAX; ++f) {
    if (f < F) {
      const half_t* vp = v + (((long long)kvb * F + f) * S + s) * ldv + hh * 64;
      // SDPA rounds the probabilities to the compute dtype before P.V
      const float p = (float)(half_t)(sc[f] * inv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        half8_t vv = *(const half8_t*)(vp + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[i * 8 + e] = fmaf(p, (float)vv[e], o[i * 8 + e]);
      }
    }
  }
  half_t* op = out + (((long long)b * Fq + fq) * S + s) * ldo + hh * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half8_t ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = (half_t)o[i * 8 + e];
    *(half8_t*)(op + i * 8) = ov;
  }
}

extern "C" int lkgd_attn_temporal(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                  void* out, int32_t ldo, int32_t B, int32_t Fq, int32_t F, int32_t S, int32_t heads,
                                  const int32_t* kv_b_map, float scale, lkgd_stream_t stream) {
  if (!q || !k || !v || !out) return LKGD_E_NULL;
  if (B <= 0 || F <= 0 || F > 32 || Fq <= 0 || Fq > F || S <= 0 || heads <= 0 || B > 65535) return LKGD_E_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return LKGD_E_ALIGN;
  if (ldq < heads * 64 || ldk < heads * 64 || ldv < heads * 64 || ldo < heads * 64) return LKGD_E_SHAPE;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) return LKGD_E_ALIGN;
  long long npairs = (long long)S * heads;
  long long nblk = (npairs + 15) / 16;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  const float c = scale * 1.4426950408889634f;
  if (F <= 16)
    hipLaunchKernelGGL(attn_temporal_kernel<16>, dim3((unsigned)nblk, B), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, Fq, F, S,
                       heads, kv_b_map, c);
  else
    hipLaunchKernelGGL(attn_temporal_kernel<32>, dim3((unsigned)nblk, B), dim3(512), 0, (hipStream_t)stream,
                       (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, Fq, F, S,
                       heads, kv_b_map, c);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
