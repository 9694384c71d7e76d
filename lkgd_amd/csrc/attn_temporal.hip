// Temporal self-attention over F <= 32 frames, head_dim 64 (include/lkgd_hip.h section 5).
//
// HBM-bound (14 keys per query: arithmetic intensity ~7 flop/B), so no MFMA.  The [B*F,S,C] <-> [B*S,F,C] regroup of the
// reference is this kernel's index map; nothing is copied.
//
// A workgroup (8 pairs x 16 or 32 query-frame slots) owns 8 adjacent (pixel, head) pairs:
//   1. K and V of the 8 pairs x F frames go to LDS by LDS-DMA (global_load_lds_dwordx4): one instruction moves one
//      (tensor, frame) plane = 8 pairs x 128 B (adjacent pairs are adjacent 128-byte head segments), all 2F planes of
//      the workgroup are in flight at once - one HBM round trip per workgroup, no staging registers; the 16-byte chunk
//      a lane fetches is XOR-ed with its pair index (source-side swizzle) so the 8 pairs of a wave read disjoint banks;
//   2. thread (pair p, query frame fq) loads ITS query row straight into registers and computes its 64-wide output row:
//      K/V rows are shared by the query threads of a pair (LDS broadcast); scores by v_dot2_f32_f16, softmax in
//      registers (exp2 with the scale folded in), P.V in fp32;
//   3. output rows leave through the (dead) K planes in LDS and are written plane by plane, 1 KiB contiguous per wave
//      instruction (stored directly, a wave instruction was 64 x 16 bytes at a 128-byte stride: 5-7 % slower at the 72x128 and
//      36x64 levels).  Staging the query planes through LDS as well costs two workgroups per CU and is slower (214 vs 171 us).
// 28 KiB of LDS per workgroup at F = 14 -> five workgroups per CU keep loads, arithmetic and stores of different
// workgroups overlapped.
#include "common.h"

#define TP 8                 // pairs per workgroup
#define TROW 128             // bytes per (frame, pair) row in LDS (chunk c of pair p lives in 16-byte slot c ^ p)

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
  constexpr int plane = TP * TROW;             // bytes per (tensor, frame) plane: one LDS-DMA instruction
  char* sk = smem;                             // [F][TP][128]
  char* sv = sk + F * plane;
  const int pp = t & (TP - 1), fq = t >> 3;

  // ---- 1a. this thread's query row -> registers (issued first: it is consumed first)
  long long my_pid = pid0 + pp;
  const bool pair_ok = my_pid < npairs;
  if (!pair_ok) my_pid = npairs - 1;
  const int my_s = (int)(my_pid / heads), my_h = (int)(my_pid - (long long)my_s * heads);
  half8_t qv[8];
  {
    const int fqc = fq < Fq ? fq : Fq - 1;
    const half_t* qp = q + (((long long)b * Fq + fqc) * S + my_s) * ldq + my_h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = *(const half8_t*)(qp + i * 8);
  }
  // ---- 1b. K / V planes by LDS-DMA: wave w takes planes w, w + nw, ...; lane = (pair lp, slot ls) fetches chunk ls ^ lp
  {
    const int lane = t & 63, w = t >> 6;
    constexpr int nw = NT / 64;
    const int lp = lane >> 3, ls = lane & 7;
    long long pid = pid0 + lp;
    if (pid >= npairs) pid = npairs - 1;
    const int s = (int)(pid / heads), hh = (int)(pid - (long long)s * heads);
    const long long koff = (long long)s * ldk + hh * 64 + ((ls ^ lp) << 3);
    const long long voff = (long long)s * ldv + hh * 64 + ((ls ^ lp) << 3);
    for (int j = w; j < 2 * F; j += nw) {
      const int f = j < F ? j : j - F;
      const half_t* src = j < F ? k + ((long long)kvb * F + f) * S * ldk + koff
                                : v + ((long long)kvb * F + f) * S * ldv + voff;
      glds16(src, smem + j * plane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 2. one thread = one (pair, query frame) row
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
  if (fq < Fq) {
    float sc[FMAX];
    float mx = -1e30f;
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
      sc[f] = -1e30f;
      if (f < F) {
        const char* kp = sk + f * plane + pp * TROW;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          half8_t kv = *(const half8_t*)(kp + ((i ^ pp) << 4));
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
    // P.V two frames at a time: o[d] += dot2((v[f][d], v[f+1][d]), (p[f], p[f+1])) - one pack + one v_dot2_f32_f16 per
    // frame pair instead of two converts + two FMAs (fp16 products are exact in fp32, the accumulator stays fp32)
#pragma unroll
    for (int f = 0; f < FMAX; f += 2) {
      if (f < F) {
        const bool two = f + 1 < F;
        const char* vp0 = sv + f * plane + pp * TROW;
        const char* vp1 = two ? vp0 + plane : vp0;
        // SDPA rounds the probabilities to the compute dtype before P.V
        const half2_t p2 = {(half_t)(sc[f] * inv), two ? (half_t)(sc[f + 1] * inv) : (half_t)0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const half8_t v0 = *(const half8_t*)(vp0 + ((i ^ pp) << 4));
          const half8_t v1 = *(const half8_t*)(vp1 + ((i ^ pp) << 4));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const half2_t vv = {v0[e], v1[e]};
            o[i * 8 + e] = __builtin_amdgcn_fdot2(vv, p2, o[i * 8 + e], false);
          }
        }
      }
    }
  }
  // ---- 3. output rows through LDS.  A thread owns one 128-byte row; stored directly, a wave instruction would write 64
  //      16-byte pieces at a 128-byte stride (a CU sustains well under 16 GB/s of such stores, tools/micro/store_bw.hip).
  //      The K planes are dead once every thread has its scores: each thread parks its row there (same XOR swizzle as K),
  //      and the waves write whole (frame, 8 pairs) planes - 1 KiB contiguous per instruction when the rows are.
  __syncthreads();
  if (fq < Fq) {
    char* orow = sk + fq * plane + pp * TROW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      half8_t ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (half_t)o[i * 8 + e];
      *(half8_t*)(orow + ((i ^ pp) << 4)) = ov;
    }
  }
  __syncthreads();
  {
    const int lane = t & 63, w = t >> 6;
    constexpr int nw = NT / 64;
    const int lp = lane >> 3, ls = lane & 7;
    const long long pid = pid0 + lp;
    if (pid < npairs) {
      const int s = (int)(pid / heads), hh = (int)(pid - (long long)s * heads);
      const long long ooff = (long long)s * ldo + hh * 64 + ((ls ^ lp) << 3);
      for (int j = w; j < Fq; j += nw)
        *(half8_t*)(out + ((long long)b * Fq + j) * S * ldo + ooff) = *(const half8_t*)(sk + j * plane + lane * 16);
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
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)attn_temporal_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * 32 * TP * TROW) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  if (F <= 16)
    hipLaunchKernelGGL(attn_temporal_kernel<16>, dim3((unsigned)nblk, B), dim3(16 * TP), 2 * F * TP * TROW,
                       (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv,
                       (half_t*)out, ldo, Fq, F, S, heads, kv_b_map, c);
  else
    hipLaunchKernelGGL(attn_temporal_kernel<32>, dim3((unsigned)nblk, B), dim3(32 * TP), 2 * F * TP * TROW,
                       (hipStream_t)stream, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv,
                       (half_t*)out, ldo, Fq, F, S, heads, kv_b_map, c);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
