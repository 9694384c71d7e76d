// Fused temporal-attention front (include/lkgd_hip.h section 5b): LayerNorm + Q|K|V projection + attention over the F frames of
// every pixel in ONE kernel - the first half of SURVEY.md finding 7 / K6 (witness /root/reference/patch/patch.py:582-686: the
// [B*F,S,C] -> [B*S,F,C] regroup :592-597, norm1 :610, attn1 :660-661).  Unfused, a temporal block at the 72x128 level writes and
// re-reads the normalised tokens (330 MB) and the Q|K|V rows (990 MB) to run 0.04 TFLOP of 14x14 attention; here both stay in
// registers and the matrix work IS the projection (AI ~ 400 flop/B instead of 7).
//
// C = 320 channels (K = 320), heads of 64, F <= 16 frames.  A workgroup (8 waves) owns 16 adjacent pixels of one clip; a wave
// owns TWO pixels x (up to) 16 frame rows = two MFMA token fragments: all F frames of a pixel sit in one wave, so the
// attention needs no exchange between waves.
//   * token rows: loaded once (16-byte loads, row = (b*F + f)*HW + pixel), LayerNorm-ed in registers (each lane holds a
//     quarter of two rows; two shuffles per statistic) and parked as MFMA operand fragments in a[32:111];
//   * weights: 15 chunks per panel (5 heads x q, k, v) of 64 rows x K = 40 KiB, fragment-major in HBM (packed once), streamed
//     through a 3-stage LDS ring by LDS-DMA, one barrier per chunk; the chunk loop is generated asm (tools/gen_tfront_asm.py);
//   * q and k are computed "swapped" (acc[d][token]: a lane owns a token, 4 head channels per fragment), v "direct"
//     (acc[token][d]: a lane owns a head channel, 4 tokens per fragment) - the same token registers with the operand order
//     exchanged.  Packed to fp16 these accumulators ARE the operands of the attention MFMAs:
//         S^T[key][query] = K . Q^T   A = k (lane = key),  B = q (lane = query);  the contraction order over the 64 head
//                                     channels is the same permutation on both sides
//         O[d][query]     = V^T . P   A = v (lane = d, k-slots = its 4 tokens + 4 zeros),  B = P (lane = query, k-slots =
//                                     its 4 keys + 4 zeros)
//     softmax over the keys of a query = 4 registers + the 4 lanes lq (two shuffles); keys >= F are masked;
//   * the head's 64 output channels of the wave's 32 rows leave through a wave-private LDS patch as 16-byte pieces of
//     128-byte row segments.
#include "common.h"

// timing-experiment knobs (tools/micro/tfront_knobs.sh; results are wrong with any of them): never defined in the product build
#ifdef TF_X_NOMFMA
#define TF_MM(X) ""
#else
#define TF_MM(X) X
#endif
#ifdef TF_X_NOREAD
#define TF_RD(X) ""
#else
#define TF_RD(X) X
#endif
#include "attn_tfront_kloop.inc"

#define TF_NT 512
#define TF_WAVES 8
#define TF_C 320
#define TF_CHUNK_BYTES 40960          // 4 fragments x 10 K-steps x 1 KiB
#define TF_STAGES 3
#define TF_PATCH_PITCH 144            // bytes per staged output row (128 + 16)
#define TF_PATCH (32 * TF_PATCH_PITCH)
#define TF_BIAS_BYTES (3 * TF_C * 4)   // the projection bias, staged once (a global load at its use would expose an L2 round trip per chunk)
#define TF_LDS (TF_STAGES * TF_CHUNK_BYTES + TF_WAVES * TF_PATCH + TF_BIAS_BYTES)

template <int REG>
__device__ __forceinline__ unsigned tf_agpr_read() {
  unsigned v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(REG));
  return v;
}
template <int REG>
__device__ __forceinline__ void tf_agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v), "i"(REG));
}
template <int BASE>
__device__ __forceinline__ float4_t tf_read_acc() {
  float a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\t"
               "v_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
               : "i"(BASE));
  return (float4_t){a, b, c, d};
}

template <int V> struct TfIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void tf_static_for(F&& f, TfIC<Is>...) { (f(TfIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void tf_for4(F&& f) { tf_static_for(f, TfIC<0>{}, TfIC<1>{}, TfIC<2>{}, TfIC<3>{}); }
template <class F> __device__ __forceinline__ void tf_for10(F&& f) {
  tf_static_for(f, TfIC<0>{}, TfIC<1>{}, TfIC<2>{}, TfIC<3>{}, TfIC<4>{}, TfIC<5>{}, TfIC<6>{}, TfIC<7>{}, TfIC<8>{}, TfIC<9>{});
}

// the 80 token-fragment loads of a panel: x[j][ks] = a[32 + (j*10 + ks)*4 ..] <- 16 bytes at row(j) + ks*64 (+ lq*16 in the pointers)
__device__ __forceinline__ void tf_load_x(const half_t* p0, const half_t* p1) {
  asm volatile(TFRONT_LOADX_ASM : : "v"(p0), "v"(p1) : "memory", TFRONT_AGPR_CLOBBERS);
}

template <bool DIRECT>
__device__ __forceinline__ void tf_chunk(int wl) {
  half8_t f0, f1, f2, f3, f4, f5;
  if (DIRECT)
    asm volatile(TFRONT_CHUNK_ASM_DIRECT : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5) : "v"(wl)
                 : "memory", TFRONT_AGPR_CLOBBERS);
  else
    asm volatile(TFRONT_CHUNK_ASM_SWAPPED : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5) : "v"(wl)
                 : "memory", TFRONT_AGPR_CLOBBERS);
}

// The attention MFMAs: compiler-generated MFMAs of a kernel whose inline asm uses AGPRs get AGPR destinations (a[0:15] here,
// checked in the ISA) - on top of the chunk accumulators a[0:31].  Both call sites therefore read every accumulator of the
// chunk BEFORE their first attention MFMA; the token fragments a[32:111] are never touched.
__device__ __forceinline__ float4_t tf_mfma(half8_t a, half8_t b) {
  const float4_t z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ float4_t tf_mfma_acc(half8_t a, half8_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float tf_quad_sum(float v) {      // over the four lanes lq of a token (lanes l15 + 16*lq)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float tf_quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

__global__ __launch_bounds__(TF_NT, 2) __attribute__((amdgpu_num_vgpr(144))) void tattn_front_kernel(
    const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wpack, const float* __restrict__ bqkv, half_t* __restrict__ out,
    int ldo, int F, int HW, int heads, int npanels, int panels_per_clip, float eps, float qscale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  char* patch = smem + TF_STAGES * TF_CHUNK_BYTES + w * TF_PATCH;
  float* bias_l = (float*)(smem + TF_STAGES * TF_CHUNK_BYTES + TF_WAVES * TF_PATCH);
  const int nchunk = heads * 3;
  const int my_panels = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_panels <= 0) return;
  const long long total = (long long)my_panels * nchunk;        // chunks in this workgroup's stream

  // the weight stream is the same 15 chunks for every panel; thread t copies 16 bytes x 5 per chunk (lane-linear LDS-DMA)
  auto issue_chunk = [&](long long g) {
    const int c = (int)(g % nchunk);
    const char* src = (const char*)wpack + (long long)c * TF_CHUNK_BYTES + t * 16;
    char* dst = smem + (int)(g % TF_STAGES) * TF_CHUNK_BYTES + w * 1024;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
#ifndef TF_X_NODMA
      glds16(src + r * 8192, dst + r * 8192);
#endif
    }
  };
  issue_chunk(0);
  if (total > 1) issue_chunk(1);
  for (int i = t; i < 3 * TF_C; i += TF_NT) bias_l[i] = bqkv ? bqkv[i] : 0.f;
  __syncthreads();

  const int fcl = l15 < F ? l15 : F - 1;            // frame of this lane's token rows (rows >= F repeat the last frame: masked / not stored)
  half8_t qp[2][2], kp[2][2];                        // head channels x tokens, fp16 MFMA operands: [k-step of 32 channels][token fragment]
  long long g = 0;
#ifdef TF_X_STAMPS      // diagnostic build (tools/micro/tfront_stamps.py): s_memtime sums per wave, written over the first output rows
  long long st_sync = 0, st_mm = 0, st_e[3] = {0, 0, 0}, st_pro = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
#define TF_STAMP(X) X = __builtin_amdgcn_s_memtime()
#else
#define TF_STAMP(X)
#endif
#pragma unroll 1
  for (int pi = 0; pi < my_panels; ++pi) {
    TF_STAMP(st_a);
    const int panel = (int)blockIdx.x + pi * (int)gridDim.x;
    const int b = panel / panels_per_clip;
    const int p0 = (panel - b * panels_per_clip) * 16 + 2 * w;
    // this lane's four output rows (piece q = lane + 64 kk -> row q >> 3 = 8 kk + lane / 8, column piece q & 7) at head 0
    half_t* dstp[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int row = 8 * kk + (lane >> 3), j = row >> 4, f = row & 15;
      dstp[kk] = out + ((long long)(b * F + (f < F ? f : F - 1)) * HW + p0 + j) * ldo + (lane & 7) * 8;
    }
    // ---- token rows of the wave's two pixels -> a[32:111]
    {
      const half_t* r0 = x + ((long long)(b * F + fcl) * HW + p0) * ldx + lq * 8;
      tf_load_x(r0, r0 + ldx);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // LayerNorm of the two rows this lane holds a quarter of (affine folded into the weights)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float s = 0.f, q2 = 0.f;
        // (j is a runtime-unrolled index: the AGPR numbers must be immediates, so both fragments are written out)
        if (j == 0) {
          tf_for10([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            tf_for4([&](auto rc) {
              constexpr int r = decltype(rc)::value;
              const half2_t u = __builtin_bit_cast(half2_t, tf_agpr_read<32 + ks * 4 + r>());
              const half2_t one = {(half_t)1.f, (half_t)1.f};
              s = __builtin_amdgcn_fdot2(u, one, s, false);
              q2 = __builtin_amdgcn_fdot2(u, u, q2, false);
            });
          });
        } else {
          tf_for10([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            tf_for4([&](auto rc) {
              constexpr int r = decltype(rc)::value;
              const half2_t u = __builtin_bit_cast(half2_t, tf_agpr_read<72 + ks * 4 + r>());
              const half2_t one = {(half_t)1.f, (half_t)1.f};
              s = __builtin_amdgcn_fdot2(u, one, s, false);
              q2 = __builtin_amdgcn_fdot2(u, u, q2, false);
            });
          });
        }
        s = tf_quad_sum(s);
        q2 = tf_quad_sum(q2);
        const float mean = s * (1.0f / TF_C);
        float var = q2 * (1.0f / TF_C) - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = __builtin_amdgcn_rsqf(var + eps);
        const float nm = -mean * rstd;
        auto norm = [&](auto regc) {
          constexpr int reg = decltype(regc)::value;
          const half2_t u = __builtin_bit_cast(half2_t, tf_agpr_read<reg>());
          const half2_t o = {(half_t)fmaf((float)u.x, rstd, nm), (half_t)fmaf((float)u.y, rstd, nm)};
          tf_agpr_write<reg>(__builtin_bit_cast(unsigned, o));
        };
        if (j == 0) {
          tf_for10([&](auto kc) { tf_for4([&](auto rc) { norm(TfIC<32 + decltype(kc)::value * 4 + decltype(rc)::value>{}); }); });
        } else {
          tf_for10([&](auto kc) { tf_for4([&](auto rc) { norm(TfIC<72 + decltype(kc)::value * 4 + decltype(rc)::value>{}); }); });
        }
      }
    }
#ifdef TF_X_STAMPS
    st_b = __builtin_amdgcn_s_memtime(); st_pro += st_b - st_a;
#endif
#pragma unroll 1
    for (int h = 0; h < heads; ++h) {
      half8_t pp[2];                                  // probabilities of the two pixels: this lane's 4 keys + 4 zero k-slots
#pragma unroll 1
      for (int which = 0; which < 3; ++which, ++g) {
        // chunk g has landed (this thread's loads; the barrier publishes everyone's); every wave is done with chunk g-1,
        // whose stage takes chunk g+2.  Issued after chunk g's loads (at the top of iteration g-2): the four row-piece stores
        // of iteration g-2 when that was a v chunk (this is a k chunk), chunk g+1's five loads, the four stores of iteration
        // g-1 when that was a v chunk (this is a q chunk) - 9 operations may stay in flight at a q / k chunk, 5 at a v chunk.
        // Only the stores with a live lane are counted (F <= 8 leaves the two that hold frames 8.. without one).
        TF_STAMP(st_a);
        if (g + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stream's last chunk: nothing behind it
        else if (which == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (F > 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");    // (first head of a panel: the panel start drained everything)
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g + 2 < total) issue_chunk(g + 2);
        const int wl = (int)(g % TF_STAGES) * TF_CHUNK_BYTES + lane * 16;
        const float* bias = bias_l + which * TF_C + h * 64;
#ifdef TF_X_STAMPS
        st_b = __builtin_amdgcn_s_memtime(); st_sync += st_b - st_a;
        if (which < 2) tf_chunk<false>(wl); else tf_chunk<true>(wl);
        st_a = __builtin_amdgcn_s_memtime(); st_mm += st_a - st_b;
#endif
#ifdef TF_X_NOEPI
        if (which < 2) tf_chunk<false>(wl); else tf_chunk<true>(wl);
        if (lane < 0)          // never true: the epilogues stay compiled, nothing runs
#endif
        if (which < 2) {
#if !defined(TF_X_NOEPI) && !defined(TF_X_STAMPS)
          tf_chunk<false>(wl);
#endif
          // acc[d][token] + bias[d] -> fp16 (q also times softmax scale * log2 e)
          const float sc = which == 0 ? qscale : 1.0f;
          tf_for4([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float4_t bb = *(const float4_t*)(bias + i * 16 + 4 * lq);
            tf_static_for([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              const float4_t v = (tf_read_acc<(2 * i + j) * 4>() + bb) * sc;
              const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              // k-slots of the S^T MFMAs: fragments 2s, 2s+1 of a lane form one 8-slot operand
              half8_t& dst = which == 0 ? qp[i >> 1][j] : kp[i >> 1][j];
              if ((i & 1) == 0) { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
              else { dst[4] = o[0]; dst[5] = o[1]; dst[6] = o[2]; dst[7] = o[3]; }
            }, TfIC<0>{}, TfIC<1>{});
          });
          if (which == 1) {
            // ---- S^T = K . Q^T per pixel (two k-steps of 32 head channels), softmax over the keys
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              float4_t s = tf_mfma(kp[0][j], qp[0][j]);
              s = tf_mfma_acc(kp[1][j], qp[1][j], s);
              // lane (query l15, lq) holds keys 4lq .. 4lq+3
#pragma unroll
              for (int r = 0; r < 4; ++r) if (4 * lq + r >= F) s[r] = -1e30f;
              float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
              m = tf_quad_max(m);
              float4_t p;
#pragma unroll
              for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - m);
              float l = (p[0] + p[1]) + (p[2] + p[3]);
              l = tf_quad_sum(l);
              const float inv = 1.0f / l;
              half8_t o;
#pragma unroll
              for (int r = 0; r < 4; ++r) { o[r] = (half_t)(p[r] * inv); o[4 + r] = (half_t)0.f; }
              pp[j] = o;
            }
          }
        } else {
#if !defined(TF_X_NOEPI) && !defined(TF_X_STAMPS)
          tf_chunk<true>(wl);
#endif
          // acc[token][d] (lane = head channel d = 16i + l15, 4 tokens 4lq..) + bias[d] -> fp16 k-slots of O = V^T . P
          // (every accumulator is read before the first attention MFMA below: see tf_mfma)
          half8_t vp[4][2];
          tf_for4([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float bb = bias[i * 16 + l15];
            const float4_t v0 = tf_read_acc<(2 * i) * 4>() + bb, v1 = tf_read_acc<(2 * i + 1) * 4>() + bb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              vp[i][0][r] = (half_t)v0[r]; vp[i][0][4 + r] = (half_t)0.f;
              vp[i][1][r] = (half_t)v1[r]; vp[i][1][4 + r] = (half_t)0.f;
            }
          });
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float4_t o = tf_mfma(vp[i][j], pp[j]);      // O[d = 16i + 4lq..][query l15]
              const half4_t hv = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
              *(half4_t*)(patch + (j * 16 + l15) * TF_PATCH_PITCH + (i * 16 + 4 * lq) * 2) = hv;
            }
          }
          // the wave's 32 rows x 128 bytes as 16-byte pieces: piece q -> row q >> 3, column piece q & 7
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int q = lane + 64 * kk;
            const int row = q >> 3, cc = q & 7;
            const half8_t v = *(const half8_t*)(patch + row * TF_PATCH_PITCH + cc * 16);
            const int f = row & 15;
            half_t* dst = dstp[kk] + h * 64;
            unsigned long long sv;
            asm volatile("v_cmp_gt_i32 vcc, %3, %4\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dwordx4 %1, %2, off\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                         : "=&s"(sv) : "v"(dst), "v"(v), "s"(F), "v"(f) : "vcc", "memory");
          }
        }
#ifdef TF_X_STAMPS
        st_e[which] += __builtin_amdgcn_s_memtime() - st_a;
#endif
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TF_X_STAMPS
  if (lane == 0) {
    long long* dbg = (long long*)out + ((long long)blockIdx.x * TF_WAVES + w) * 8;
    dbg[0] = st_pro; dbg[1] = st_sync; dbg[2] = st_mm; dbg[3] = st_e[0]; dbg[4] = st_e[1]; dbg[5] = st_e[2];
    dbg[6] = __builtin_amdgcn_s_memtime() - st_t0; dbg[7] = my_panels;
  }
#endif
}

extern "C" int lkgd_tattn_front(const void* x, int32_t ldx, const void* wpack, const float* bqkv, void* out, int32_t ldo, int32_t B,
                                int32_t F, int32_t HW, int32_t heads, float eps, float scale, lkgd_stream_t stream) {
  if (!x || !wpack || !out) return LKGD_E_NULL;
  if (B <= 0 || F <= 0 || F > 16 || HW <= 0 || HW % 16 || heads * 64 != TF_C) return LKGD_E_SHAPE;
  if (ldx % 8 || ldo % 8 || ldx < TF_C || ldo < TF_C || !aligned16(x) || !aligned16(wpack) || !aligned16(out)) return LKGD_E_ALIGN;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)tattn_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const int cus = lkgd_cu_count();      // cached per device (common.h)
  const int panels_per_clip = HW / 16;
  const long long npanels = (long long)B * panels_per_clip;
  if (npanels > 0x7fffffffLL) return LKGD_E_SHAPE;
  const int grid = npanels < cus ? (int)npanels : cus;
  hipLaunchKernelGGL(tattn_front_kernel, dim3(grid), dim3(TF_NT), TF_LDS, (hipStream_t)stream, (const half_t*)x, ldx,
                     (const half_t*)wpack, bqkv, (half_t*)out, ldo, F, HW, heads, (int)npanels, panels_per_clip, eps,
                     scale * 1.4426950408889634f);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
