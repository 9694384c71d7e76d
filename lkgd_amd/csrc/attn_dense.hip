// Dense self-attention for SHORT sequences with any head_dim <= 128 (include/lkgd_hip.h section 16): the attention of the
// CLIP-ViT-H image encoder in front of the loop (257 tokens, 16 heads of 80 channels - the head_dim-64 flash kernel does not
// fit; /root/reference/pipeline/pipeline_stable_video_diffusion_trans.py:164-203 runs `self.image_encoder(image).image_embeds`
// once per clip).  10.8 GFLOP per encoded image in 32 launches: a boundary stage, written for exactness and brevity, not for the
// matrix pipe - fp16 operands, every product and sum in fp32.
//
// A workgroup = one (image, head) and AD_QPB query rows; the head's K and V rows sit in LDS (row pitch head_dim + 2 halfs: an
// odd number of dwords, so the lanes of a wave - one key each - read their K rows without bank conflicts).  A wave takes one
// query at a time: scores with one key per lane (v_dot2_f32_f16 over the head channels), wave-wide max / sum, the
// unnormalised probabilities parked in LDS, then O = P . V with one channel PAIR per lane.
#include "common.h"

#define AD_NT 256
#define AD_WAVES 4
#define AD_QPB 16             // query rows per workgroup
#define AD_MAXD 128

__device__ __forceinline__ float ad_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float ad_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(AD_NT) void attn_dense_kernel(const half_t* __restrict__ q, int ldq, const half_t* __restrict__ k, int ldk,
                                                           const half_t* __restrict__ v, int ldv, half_t* __restrict__ out, int ldo,
                                                           int S, int heads, int D, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int bh = blockIdx.x, n = bh / heads, h = bh - n * heads;
  const int D2 = D >> 1, P2 = D2 + 1;                   // dwords per row / LDS pitch in dwords
  half2_t* sk = (half2_t*)smem;                         // [S][P2]
  half2_t* sv = sk + (long long)S * P2;                 // [S][P2]
  float* sp = (float*)(sv + (long long)S * P2) + w * S; // [waves][S] probabilities of the wave's query
  half2_t* sq = (half2_t*)((float*)(sv + (long long)S * P2) + AD_WAVES * S) + w * (AD_MAXD / 2);   // [waves][64] its query row
  const half_t* kb = k + (long long)n * S * ldk + h * D;
  const half_t* vb = v + (long long)n * S * ldv + h * D;
  const int c8 = D >> 3;                                // 16-byte chunks per row
  for (int c = t; c < S * c8; c += AD_NT) {
    const int r = c / c8, cc = c - r * c8;
    const half8_t kx = *(const half8_t*)(kb + (long long)r * ldk + cc * 8);
    const half8_t vx = *(const half8_t*)(vb + (long long)r * ldv + cc * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sk[r * P2 + cc * 4 + e] = (half2_t){kx[2 * e], kx[2 * e + 1]};
      sv[r * P2 + cc * 4 + e] = (half2_t){vx[2 * e], vx[2 * e + 1]};
    }
  }
  __syncthreads();
  const int q0 = blockIdx.y * AD_QPB;
  for (int qi = w; qi < AD_QPB; qi += AD_WAVES) {
    const int qr = q0 + qi;
    if (qr >= S) break;                                  // wave-uniform
    const half_t* qp = q + ((long long)n * S + qr) * ldq + h * D;
    if (lane < D2) sq[lane] = *(const half2_t*)(qp + 2 * lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the wave's own LDS writes are visible to its reads below
    // ---- scores: key = lane, lane + 64, ...
    float mx = -INFINITY;
    for (int key = lane; key < S; key += 64) {
      const half2_t* kr = sk + key * P2;
      float s = 0.f;
      for (int c = 0; c < D2; ++c) s = __builtin_amdgcn_fdot2(kr[c], sq[c], s, false);
      s *= scale;
      sp[key] = s;
      mx = fmaxf(mx, s);
    }
    mx = ad_wave_max(mx);
    float l = 0.f;
    for (int key = lane; key < S; key += 64) {
      const float p = __expf(sp[key] - mx);
      sp[key] = p;
      l += p;
    }
    l = ad_wave_sum(l);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- O = P . V: lane = channel pair
    if (lane < D2) {
      float a0 = 0.f, a1 = 0.f;
      for (int key = 0; key < S; ++key) {
        const half2_t vv = sv[key * P2 + lane];
        const float p = sp[key];
        a0 = fmaf(p, (float)vv.x, a0);
        a1 = fmaf(p, (float)vv.y, a1);
      }
      const float inv = 1.0f / l;
      *(half2_t*)(out + ((long long)n * S + qr) * ldo + h * D + 2 * lane) = (half2_t){(half_t)(a0 * inv), (half_t)(a1 * inv)};
    }
  }
}

static size_t attn_dense_lds(int S, int D) {
  return (size_t)2 * S * (D / 2 + 1) * 4 + (size_t)AD_WAVES * S * 4 + (size_t)AD_WAVES * (AD_MAXD / 2) * 4;
}

extern "C" int lkgd_attn_dense(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                               int32_t ldo, int32_t nbatch, int32_t S, int32_t heads, int32_t head_dim, float scale,
                               lkgd_stream_t stream) {
  if (!q || !k || !v || !out) return LKGD_E_NULL;
  if (nbatch <= 0 || S <= 0 || heads <= 0 || head_dim <= 0 || head_dim > AD_MAXD || head_dim % 8) return LKGD_E_SHAPE;
  const size_t lds = attn_dense_lds(S, head_dim);
  if (lds > 160 * 1024) return LKGD_E_SHAPE;             // K and V of one head must fit the CU's LDS
  const int width = heads * head_dim;
  if (ldq < width || ldk < width || ldv < width || ldo < width) return LKGD_E_SHAPE;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 2 || !aligned16(q) || !aligned16(k) || !aligned16(v) || ((uintptr_t)out & 3))
    return LKGD_E_ALIGN;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)attn_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const long long nbh = (long long)nbatch * heads;
  if (nbh > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(attn_dense_kernel, dim3((unsigned)nbh, (unsigned)((S + AD_QPB - 1) / AD_QPB)), dim3(AD_NT), lds,
                     (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, S,
                     heads, head_dim, scale);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
