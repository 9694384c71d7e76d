// Spatial self-attention, head_dim 64, the SOFTWARE-PIPELINED program (include/lkgd_hip.h section 4; round 4).
//
// Same arithmetic as attn_spatial.hip (swapped product S^T = K.Q^T, the reference maximum subtracted by a bias k-step of the
// matrix pipe, exp2 straight off the accumulators, the fp16 probabilities ARE the B operand of O^T += V^T.P^T, V^T by
// ds_read_b64_tr_b16), but the main loop is ONE generated inline-asm statement (tools/gen_attn_asm.py ->
// attn_spatial_pipe.inc): a wave owns two 32-query tiles and every MFMA issue slot of one tile carries the exponentials,
// row-sum adds and conversions of the other, so a single wave keeps the SIMD's vector-issue port busy instead of leaving the
// overlap of QK^T / softmax / P.V to the chance interleaving of four waves (44.6 % matrix-pipe busy, profiles/
// r02_pmc_attn_spatial.txt).  Workgroup = 8 waves x 64 queries = 512 queries; K / V stages of 128 keys in a three-buffer LDS
// ring filled by LDS-DMA one stage ahead, one barrier per stage; 2 waves per SIMD (148 arch + 108 accumulation registers).
// S % 128 != 0 (CogVideoX: 17 776 tokens) runs the MASKED form of the same statement: the last 128-key stage is loaded from
// keys [S - 128, S) - it overlaps the stage before it by dup = 128 - S % 128 keys - and those duplicates are masked through a
// second k-slot of the bias k-step (1.0 on the key side for exactly those rows, -30000 on the query side), so nothing is read
// beyond the S rows of a batch entry.  Short sequences stay on attn_spatial.hip.
#include "common.h"
#include "attn_spatial_pipe.inc"
#include "attn_spatial_pipe_masked.inc"     // the same program for S % 128 != 0 (ATTN_GEN_OPT=w2+mask)

#define AP_NW 8
#define AP_STAGE 32768
#define AP_LDS (3 * AP_STAGE)

template <int REG>
__device__ __forceinline__ float ap_agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(REG));
  return v;
}
template <int REG>
__device__ __forceinline__ void ap_agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v), "i"(REG));
}
template <int V> struct ApIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void ap_static_for(F&& f, ApIC<Is>...) { (f(ApIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void ap_for4(F&& f) { ap_static_for(f, ApIC<0>{}, ApIC<1>{}, ApIC<2>{}, ApIC<3>{}); }
template <class F> __device__ __forceinline__ void ap_for2(F&& f) { ap_static_for(f, ApIC<0>{}, ApIC<1>{}); }

// Q fragments of one tile -> a[BASE .. BASE+15]: lane holds Q[qrow][16*ks + 8*h + 0..7] * scale*log2e (B operand of S^T = K.Q^T)
template <int BASE>
__device__ __forceinline__ void ap_load_q(const half_t* qp, float scale_log2e) {
  ap_for4([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    const half8_t raw = *(const half8_t*)(qp + ks * 16);
    half8_t sc;
#pragma unroll
    for (int e = 0; e < 8; ++e) sc[e] = (half_t)((float)raw[e] * scale_log2e);
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    const uint4_t u = __builtin_bit_cast(uint4_t, sc);
    ap_agpr_write<BASE + 4 * ks + 0>(u[0]);
    ap_agpr_write<BASE + 4 * ks + 1>(u[1]);
    ap_agpr_write<BASE + 4 * ks + 2>(u[2]);
    ap_agpr_write<BASE + 4 * ks + 3>(u[3]);
  });
}

// normalise and store one tile: lane owns query row qrow, d = 32*df + 8*g + 4*h + e  <-  a[OBASE + 16*df + 4*g + e]
template <int OBASE>
__device__ __forceinline__ void ap_store_tile(half_t* op, float inv, bool valid) {
  ap_for2([&](auto df_) {
    constexpr int df = decltype(df_)::value;
    ap_for4([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      half4_t o;
      o[0] = (half_t)(ap_agpr_read<OBASE + 16 * df + 4 * g + 0>() * inv);
      o[1] = (half_t)(ap_agpr_read<OBASE + 16 * df + 4 * g + 1>() * inv);
      o[2] = (half_t)(ap_agpr_read<OBASE + 16 * df + 4 * g + 2>() * inv);
      o[3] = (half_t)(ap_agpr_read<OBASE + 16 * df + 4 * g + 3>() * inv);
      if (valid) *(half4_t*)(op + 32 * df + 8 * g) = o;
    });
  });
}

template <bool MASKED>
__global__ __launch_bounds__(AP_NW * 64, 1) __attribute__((amdgpu_num_vgpr(ATTN_PIPE_VEND))) void attn_pipe_kernel(
    const half_t* __restrict__ q, int ldq, const half_t* __restrict__ k, int ldk, const half_t* __restrict__ v, int ldv,
    half_t* __restrict__ out, int ldo, int Sq, int S, int heads, const int* __restrict__ kvmap, float scale_log2e, int nqb,
    int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int h = lane >> 5, l31 = lane & 31;

  // XCD-aware bijective remap: blocks of one (batch, head) stay on one XCD so K/V are served from its L2
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, slot = bid >> 3;
    int qq = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
  }
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int n = bh / heads, head = bh - n * heads;
  const int kvn = kvmap ? kvmap[n] : n;

  // ---- stage 0 of K / V by LDS-DMA (this wave's four 1-KiB pieces; the statement below issues every later stage the same way):
  //      thread fills 16-byte position sc of tile row p*64 + w*8 + (lane>>3); the XOR swizzles are applied on the SOURCE side
  const half_t* kbase = k + (long long)kvn * S * ldk + head * 64;
  const half_t* vbase = v + (long long)kvn * S * ldv + head * 64;
  const int srow0 = w * 8 + (lane >> 3), sc = lane & 7;
  const int kc0 = (sc ^ ((srow0 >> 1) & 7)) * 8;
  const int vc0 = (sc ^ (((srow0 >> 1) & 1) << 2)) * 8;
  const unsigned vok = (unsigned)(srow0 * ldk + kc0) * 2u, vov = (unsigned)(srow0 * ldv + vc0) * 2u;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    char* dst = smem + (p * 64 + w * 8) * 128;
    glds16((const char*)kbase + (size_t)(p * 64) * ldk * 2 + vok, dst);
    glds16((const char*)vbase + (size_t)(p * 64) * ldv * 2 + vov, dst + 16384);
  }

  // ---- Q fragments of the wave's two 32-query tiles -> accumulation registers
  const int qrowA = qb * (AP_NW * 64) + w * 64 + l31, qrowB = qrowA + 32;
  {
    const int ra = qrowA < Sq ? qrowA : Sq - 1, rb = qrowB < Sq ? qrowB : Sq - 1;
    ap_load_q<ATTN_PIPE_QF_A>(q + ((long long)n * Sq + ra) * ldq + head * 64 + h * 8, scale_log2e);
    ap_load_q<ATTN_PIPE_QF_B>(q + ((long long)n * Sq + rb) * ldq + head * 64 + h * 8, scale_log2e);
  }

  // ---- per-lane LDS byte addresses of the fragment reads (buffer 0; the statement rotates them through the ring)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned ka0, ka1, ka2, ka3, va0, va1;
  {
    const int sw = (l31 >> 1) & 7;
    ka0 = lds0 + l31 * 128 + (((0 + h) ^ sw) << 4);
    ka1 = lds0 + l31 * 128 + (((2 + h) ^ sw) << 4);
    ka2 = lds0 + l31 * 128 + (((4 + h) ^ sw) << 4);
    ka3 = lds0 + l31 * 128 + (((6 + h) ^ sw) << 4);
    // transposed-read lane constants: 16-lane group -> (h, dgrp); lane in group i -> (row i>>2, column part i&3)
    const int i16 = lane & 15, dgrp = (lane >> 4) & 1;
    const int tr_row = 4 * h + (i16 >> 2), tr_c = dgrp * 2 + ((i16 & 3) >> 1), tr_sub = (i16 & 1) * 8;
    const int vsw = ((tr_row >> 1) & 1) << 2;
    va0 = lds0 + tr_row * 128 + (((0 + tr_c) ^ vsw) << 4) + tr_sub;
    va1 = lds0 + tr_row * 128 + (((4 + tr_c) ^ vsw) << 4) + tr_sub;
  }
  const unsigned xora = (unsigned)((lane ^ 32) * 4);
  const unsigned hmask = h == 0 ? 0xffffu : 0u;

  // ---- uniform operands: source of stage 1, its LDS destination, strides
  const int nstages = MASKED ? (S + 127) / 128 : S / 128;
  const int dup = nstages * 128 - S;                       // MASKED: keys the last stage shares with the one before it
  const int row1 = (MASKED && nstages == 2) ? S - 128 : 128;     // first key of stage 1
  const unsigned long long kp = (unsigned long long)(uintptr_t)kbase + (unsigned long long)row1 * ldk * 2;
  const unsigned long long vp = (unsigned long long)(uintptr_t)vbase + (unsigned long long)row1 * ldv * 2;
  const unsigned klo = __builtin_amdgcn_readfirstlane((unsigned)kp), khi = __builtin_amdgcn_readfirstlane((unsigned)(kp >> 32));
  const unsigned vlo = __builtin_amdgcn_readfirstlane((unsigned)vp), vhi = __builtin_amdgcn_readfirstlane((unsigned)(vp >> 32));
  const unsigned dst1 = __builtin_amdgcn_readfirstlane(lds0 + AP_STAGE + w * 1024);
  const unsigned ldsend = __builtin_amdgcn_readfirstlane(lds0 + 3 * AP_STAGE);
  const unsigned nst = __builtin_amdgcn_readfirstlane((unsigned)nstages);
  const unsigned kp1 = __builtin_amdgcn_readfirstlane((unsigned)(64 * ldk * 2)), vp1 = __builtin_amdgcn_readfirstlane((unsigned)(64 * ldv * 2));
  const unsigned kstr = __builtin_amdgcn_readfirstlane((unsigned)(128 * ldk * 2)), vstr = __builtin_amdgcn_readfirstlane((unsigned)(128 * ldv * 2));

  float la, lb;
  if constexpr (MASKED) {
    // key row m of a 32-key tile is lane & 31; the bias k-step's A operand: 1.0 in k-slot 0 (| 1.0 in k-slot 1 on duplicate rows);
    // its B operand carries -30000 in k-slot 1
    const unsigned vm = (unsigned)l31, hm32 = h == 0 ? 0xffffffffu : 0u;
    const unsigned dupk = __builtin_amdgcn_readfirstlane((unsigned)dup);
    const unsigned klast = __builtin_amdgcn_readfirstlane((unsigned)((128 - dup) * ldk * 2));
    const unsigned vlast = __builtin_amdgcn_readfirstlane((unsigned)((128 - dup) * ldv * 2));
    asm volatile(ATTN_PIPEM_ASM
                 : [la] "=&v"(la), [lb] "=&v"(lb), [ka0] "+v"(ka0), [ka1] "+v"(ka1), [ka2] "+v"(ka2), [ka3] "+v"(ka3),
                   [va0] "+v"(va0), [va1] "+v"(va1)
                 : [vok] "v"(vok), [vov] "v"(vov), [xora] "v"(xora), [hmask] "v"(hmask), [vm] "v"(vm), [hm32] "v"(hm32), [klo] "s"(klo), [khi] "s"(khi), [vlo] "s"(vlo), [vhi] "s"(vhi),
                   [dst1] "s"(dst1), [nst] "s"(nst), [kp1] "s"(kp1), [vp1] "s"(vp1), [kstr] "s"(kstr), [vstr] "s"(vstr),
                   [ldsend] "s"(ldsend), [dup] "s"(dupk), [klast] "s"(klast), [vlast] "s"(vlast)
                 : ATTN_PIPEM_CLOBBERS);
  } else {
    asm volatile(ATTN_PIPE_ASM
                 : [la] "=&v"(la), [lb] "=&v"(lb), [ka0] "+v"(ka0), [ka1] "+v"(ka1), [ka2] "+v"(ka2), [ka3] "+v"(ka3),
                   [va0] "+v"(va0), [va1] "+v"(va1)
                 : [vok] "v"(vok), [vov] "v"(vov), [xora] "v"(xora), [hmask] "v"(hmask), [klo] "s"(klo), [khi] "s"(khi),
                   [vlo] "s"(vlo), [vhi] "s"(vhi), [dst1] "s"(dst1), [nst] "s"(nst), [kp1] "s"(kp1), [vp1] "s"(vp1),
                   [kstr] "s"(kstr), [vstr] "s"(vstr), [ldsend] "s"(ldsend)
                 : ATTN_PIPE_CLOBBERS);
  }

  // ---- normalise and store.  The lane's row is re-derived from a fresh lane id so that no address register has to live
  //      across the statement (the compiler has v0..v23 there: its 14 operands and little else)
  int lane2;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
  const int h2 = lane2 >> 5, l2 = lane2 & 31;
  const float la_t = la + __shfl_xor(la, 32, 64), lb_t = lb + __shfl_xor(lb, 32, 64);
  const int qa = qb * (AP_NW * 64) + w * 64 + l2, qbb = qa + 32;
  const int ra2 = qa < Sq ? qa : Sq - 1, rb2 = qbb < Sq ? qbb : Sq - 1;
  ap_store_tile<ATTN_PIPE_O_A>(out + ((long long)n * Sq + ra2) * ldo + head * 64 + 4 * h2, 1.0f / la_t, qa < Sq);
  ap_store_tile<ATTN_PIPE_O_B>(out + ((long long)n * Sq + rb2) * ldo + head * 64 + 4 * h2, 1.0f / lb_t, qbb < Sq);
}

// called by lkgd_attn_spatial_qk (attn_spatial.hip) after its argument checks; S >= 128
int lkgd_attn_pipe_launch(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                          int32_t ldo, int32_t nbatch, int32_t Sq, int32_t S, int32_t heads, const int32_t* kv_batch_map,
                          float scale, hipStream_t stream) {
  if (S < 128) return LKGD_E_SHAPE;
  const bool masked = S % 128 != 0;        // then the last stage is keys [S - 128, S), its duplicates masked
  // the statement addresses K / V rows with 32-bit byte offsets inside a 64-row half stage
  if ((long long)64 * ldk * 2 > 0x7fffffffLL || (long long)64 * ldv * 2 > 0x7fffffffLL) return LKGD_E_SHAPE;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)attn_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, AP_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, AP_LDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const int QBLK = AP_NW * 64;
  const int nqb = (Sq + QBLK - 1) / QBLK;
  const long long nwg = (long long)nqb * nbatch * heads;
  if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
  if (masked)
    hipLaunchKernelGGL(attn_pipe_kernel<true>, dim3((unsigned)nwg), dim3(AP_NW * 64), AP_LDS, stream, (const half_t*)q, ldq,
                       (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, Sq, S, heads, kv_batch_map,
                       scale * 1.4426950408889634f, nqb, (int)nwg);
  else
    hipLaunchKernelGGL(attn_pipe_kernel<false>, dim3((unsigned)nwg), dim3(AP_NW * 64), AP_LDS, stream, (const half_t*)q, ldq,
                       (const half_t*)k, ldk, (const half_t*)v, ldv, (half_t*)out, ldo, Sq, S, heads, kv_batch_map,
                       scale * 1.4426950408889634f, nqb, (int)nwg);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
