// Temporal self-attention over F <= 32 frames, head_dim 64 (include/lkgd_hip.h section 5).
//
// HBM-bound (14 keys per query: arithmetic intensity ~7 flop/B), so no MFMA.  The [B*F,S,C] <-> [B*S,F,C] regroup of the
// reference is this kernel's index map; nothing is copied.
//
// A workgroup (8 pairs x 16 or 32 query-frame slots) owns 8 adjacent (pixel, head) pairs:
//   1. Q, K, V of the 8 pairs x F frames are staged into LDS with fully coalesced 16-byte loads (adjacent pairs are
//      adjacent 128-byte head segments) - every byte crosses HBM / L2 exactly once;
//   2. thread (pair p, query frame fq) computes its 64-wide output row from LDS: K/V rows are shared by the query threads
//      of a pair (LDS broadcast); the pair stride is padded to 144 B so the 8 pairs of a wave hit disjoint banks;
//      scores by v_dot2_f32_f16, softmax in registers (exp2 with the scale folded in), P.V in fp32;
//   3. the output rows go back through LDS and leave as coalesced 16-byte stores.
#include "common.h"

#define TP 8                 // pairs per workgroup
#define TPAD 144             // bytes per (frame, pair) row in LDS: 128 + 16 (bank spread)

template <int FMAX>
__global__ __launch_bounds__(FMAX * TP) void attn_temporal_kernel(const half_t* __restrict__ q, int ldq,
                                                                  const half_t* __restrict__ k, int ldk,
                                                                  const half_t* __restrict__ v, int ldv,
                                                                  half_t* __restrict__ out, int ldo, int Fq, int F,
                                                                  int S, int heads, const int* __restrict__ kvmap,
                                                                  float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = FMAX * TP;
  const int t = threadIdx.x;
  const int b = blockIdx.y;
  const int kvb = kvmap ? kvmap[b] : b;
  const long long npairs = (long long)S * heads;
  const long long pid0 = (long long)blockIdx.x * TP;
  constexpr int plane = TP * TPAD;             // bytes per frame
  char* sq = smem;                             // [Fq][TP][144]
  char* sk = sq + Fq * plane;                  // [F ][TP][144]
  char* sv = sk + F * plane;

  // ---- 1. coalesced staging: chunk id -> (tensor, frame, pair, 16-byte chunk); loads are issued in batches of 8 per
  //         thread before any of them is consumed, so a thread keeps 8 x 16 B in flight instead of one
  const int nq = Fq * TP * 8, nkv = F * TP * 8;
  const int nall = nq + 2 * nkv;
  for (int base = 0; base < nall; base += 8 * NT) {
    uint4 r0, r1, r2, r3, r4, r5, r6, r7;
    int d0, d1, d2, d3, d4, d5, d6, d7;
#define TLOAD(U, R, D)                                                                                   \
    {                                                                                                      \
      int id = base + (U) * NT + t;                                                                        \
      const bool ok = id < nall;                                                                           \
      if (!ok) id = nall - 1;                                                                              \
      const int tensor = id < nq ? 0 : (id < nq + nkv ? 1 : 2);                                            \
      const int r = id - (tensor == 0 ? 0 : (tensor == 1 ? nq : nq + nkv));                                \
      const int c = r & 7, pp = (r >> 3) & (TP - 1), f = r >> 6;                                           \
      long long pid = pid0 + pp;                                                                           \
      if (pid >= npairs) pid = npairs - 1;                                                                 \
      const int s = (int)(pid / heads), hh = (int)(pid - (long long)s * heads);                            \
      const half_t* src = tensor == 0 ? q + (((long long)b * Fq + f) * S + s) * ldq                        \
                          : tensor == 1 ? k + (((long long)kvb * F + f) * S + s) * ldk                     \
                                        : v + (((long long)kvb * F + f) * S + s) * ldv;                    \
      const int off = tensor == 0 ? 0 : (tensor == 1 ? Fq * plane : (Fq + F) * plane);                     \
      R = *(const uint4*)(src + hh * 64 + c * 8);                                                          \
      D = ok ? off + f * plane + pp * TPAD + c * 16 : -1;                                                  \
    }
    TLOAD(0, r0, d0) TLOAD(1, r1, d1) TLOAD(2, r2, d2) TLOAD(3, r3, d3)
    TLOAD(4, r4, d4) TLOAD(5, r5, d5) TLOAD(6, r6, d6) TLOAD(7, r7, d7)
#undef TLOAD
    if (d0 >= 0) *(uint4*)(smem + d0) = r0;
    if (d1 >= 0) *(uint4*)(smem + d1) = r1;
    if (d2 >= 0) *(uint4*)(smem + d2) = r2;
    if (d3 >= 0) *(uint4*)(smem + d3) = r3;
    if (d4 >= 0) *(uint4*)(smem + d4) = r4;
    if (d5 >= 0) *(uint4*)(smem + d5) = r5;
    if (d6 >= 0) *(uint4*)(smem + d6) = r6;
    if (d7 >= 0) *(uint4*)(smem + d7) = r7;
  }
  __syncthreads();

  // ---- 2. one thread = one (pair, query frame) row
  const int pp = t & (TP - 1), fq = t >> 3;
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
  if (fq < Fq) {
    half8_t qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = *(const half8_t*)(sq + fq * plane + pp * TPAD + i * 16);
    float sc[FMAX];
    float mx = -1e30f;
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
      sc[f] = -1e30f;
      if (f < F) {
        const char* kp = sk + f * plane + pp * TPAD;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          half8_t kv = *(const half8_t*)(kp + i * 16);
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
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
      if (f < F) {
        const char* vp = sv + f * plane + pp * TPAD;
        // SDPA rounds the probabilities to the compute dtype before P.V
        const float p = (float)(half_t)(sc[f] * inv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          half8_t vv = *(const half8_t*)(vp + i * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[i * 8 + e] = fmaf(p, (float)vv[e], o[i * 8 + e]);
        }
      }
    }
    // ---- 3. outputs through LDS (the Q region: a thread overwrites only the q row it alone has consumed)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      half8_t ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (half_t)o[i * 8 + e];
      *(half8_t*)(sq + fq * plane + pp * TPAD + i * 16) = ov;
    }
  }
  __syncthreads();
  for (int id = t; id < nq; id += NT) {
    const int c = id & 7, p2 = (id >> 3) & (TP - 1), f = id >> 6;
    const long long pid = pid0 + p2;
    if (pid < npairs) {
      const int s = (int)(pid / heads), hh = (int)(pid - (long long)s * heads);
      *(uint4*)(out + (((long long)b * Fq + f) * S + s) * ldo + hh * 64 + c * 8) =
          *(const uint4*)(sq + f * plane + p2 * TPAD + c * 16);
    }
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
  long long nblk = (npairs + TP - 1) / TP;
  if (nblk > 0x7fffffffLL) return LKGD_E_SHAPE;
  const float c = scale * 1.4426950408889634f;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_temporal_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * 32 * TP * TPAD) != hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  if (F <= 16)
    hipLaunchKernelGGL(attn_temporal_kernel<16>, dim3((unsigned)nblk, B), dim3(16 * TP), (Fq + 2 * F) * TP * TPAD,
                       (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv,
                       (half_t*)out, ldo, Fq, F, S, heads, kv_b_map, c);
  else
    hipLaunchKernelGGL(attn_temporal_kernel<32>, dim3((unsigned)nblk, B), dim3(32 * TP), (Fq + 2 * F) * TP * TPAD,
                       (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv,
                       (half_t*)out, ldo, Fq, F, S, heads, kv_b_map, c);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
