// LayerNorm + GEGLU feed-forward + output projection + residual(s) of the 72x128 level (C = 320, inner 1280) in ONE kernel
// (include/lkgd_hip.h section 1b; round 4).  Replaces, per transformer feed-forward of that level, a LayerNorm pass, the GEGLU
// GEMM [T, 2560] <- [T, 320], the 660-MB [T, 1280] intermediate and the FF-out GEMM: BasicTransformerBlock's `norm3 -> ff`
// (patch/patch.py:551-580) and TemporalBasicTransformerBlock's `norm_in -> ff_in` / `norm3 -> ff` (:599-608, :670-680).
//
// A wave owns 32 token rows as MFMA B operands (LayerNorm-ed in registers, parked in a[160:239]); the two weight matrices
// stream L2 -> LDS once per 128-token panel (a workgroup = 4 waves, one per SIMD) as a chunk stream in the order the generated
// loop consumes it (tools/gen_ff_asm.py -> ff_fused_loop.inc, packing.pack_ff_fused); hidden * gelu(gate) stays in registers
// and IS the B operand of the second product; Y^T (320 x 32 per wave) lives in a[0:159] for the whole panel.  The statement
// of one panel is generated asm; this file is the prologue (token rows, LayerNorm) and the epilogue (bias, residuals, store).
#include "common.h"
#include "ff_fused_loop.inc"

#define FF_WAVES 4
#define FF_C 320
#define FF_B2_OFF (FF_NSLOT * FF_SLOT)       // the output bias, 320 floats behind the ring
#define FF_LDS (FF_B2_OFF + FF_C * 4)

template <int REG>
__device__ __forceinline__ float ff_agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(REG));
  return v;
}
template <int REG>
__device__ __forceinline__ void ff_agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v), "i"(REG));
}
// v_permlane32_swap_b32: lanes 32..63 of `a` <-> lanes 0..31 of `b`.  Through the builtin, not inline asm: the instruction has
// wait-state requirements after a VALU write of its operands that only the compiler's hazard recogniser sees
__device__ __forceinline__ void ff_swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
template <int V> struct FfIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void ff_static_for(F&& f, FfIC<Is>...) { (f(FfIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void ff_for4(F&& f) { ff_static_for(f, FfIC<0>{}, FfIC<1>{}, FfIC<2>{}, FfIC<3>{}); }
template <class F> __device__ __forceinline__ void ff_for10(F&& f) {
  ff_static_for(f, FfIC<0>{}, FfIC<1>{}, FfIC<2>{}, FfIC<3>{}, FfIC<4>{}, FfIC<5>{}, FfIC<6>{}, FfIC<7>{}, FfIC<8>{}, FfIC<9>{});
}
template <class F> __device__ __forceinline__ void ff_for20(F&& f) {
  ff_static_for(f, FfIC<0>{}, FfIC<1>{}, FfIC<2>{}, FfIC<3>{}, FfIC<4>{}, FfIC<5>{}, FfIC<6>{}, FfIC<7>{}, FfIC<8>{}, FfIC<9>{},
                FfIC<10>{}, FfIC<11>{}, FfIC<12>{}, FfIC<13>{}, FfIC<14>{}, FfIC<15>{}, FfIC<16>{}, FfIC<17>{}, FfIC<18>{}, FfIC<19>{});
}

struct ff_params {
  const half_t* x; int ldx; long long T;
  const half_t* rowbias; int ldrb, rb_d1, rb_md;
  const char* wstream;
  const float* b2;
  float eps, s_acc, r2;
  const half_t* res2; int ldr2;
  half_t* out; int ldo;
  int npanels;
};

// PE: a row-indexed bias is added to x before the LayerNorm (and is part of the residual): ff_in.  R2: a second residual
// (the AlphaBlender form of the temporal block's last feed-forward).
template <bool PE, bool R2>
__global__ __launch_bounds__(FF_WAVES * 64, 1) __attribute__((amdgpu_num_vgpr(256))) void ff_fused_kernel(ff_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  if ((int)blockIdx.x >= p.npanels) return;

  // ---- chunks 0..2 of the stream (every panel's statement issues the chunks three ahead, across panel borders)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const char* src = p.wstream + c * FF_W1_BYTES + lane * 16;
    char* dst = smem + c * FF_SLOT;
#pragma unroll
    for (int j = 0; j < 5; ++j) glds16(src + (w + 4 * j) * 1024, dst + (w + 4 * j) * 1024);
    glds16(src + 20480, dst + 20480);
  }
  // the output bias goes to LDS once: the epilogue then needs no global load at all, so nothing in it waits (vmcnt is
  // in-order) behind the NEXT panel's token rows, which are on their way from memory while it runs
  for (int i = t; i < FF_C; i += FF_WAVES * 64) ((float*)(smem + FF_B2_OFF))[i] = p.b2[i];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned long long sp0 = (unsigned long long)(uintptr_t)p.wstream;
  const unsigned sp0lo = __builtin_amdgcn_readfirstlane((unsigned)sp0), sp0hi = __builtin_amdgcn_readfirstlane((unsigned)(sp0 >> 32));
  unsigned splo, sphi;
  {
    const unsigned long long sp = sp0 + 3ull * FF_W1_BYTES;
    splo = __builtin_amdgcn_readfirstlane((unsigned)sp);
    sphi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
  }
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)w * 1024u);
  const unsigned lds0u = __builtin_amdgcn_readfirstlane(lds0);
  const unsigned fa0 = lds0 + lane * 16, fa1 = fa0 + 2 * FF_SLOT, fa2 = fa0 + 4 * FF_SLOT;
  const unsigned vo0 = (unsigned)(w * 1024 + lane * 16), vo1 = vo0 + 4096, vo2 = vo0 + 8192, vo3 = vo0 + 12288, vo4 = vo0 + 16384;
  const unsigned vob = 20480u + lane * 16;
  const unsigned hmask = h == 0 ? 0xffffffffu : 0u;

  // the wave's 32 token rows of a panel: this lane holds channels 16 ks + 8 h + 0..7 of its row (the B operand layout)
  auto load_rows = [&](int panel, half8_t (&raw)[20], half8_t (&pv)[20], bool bias_only) {
    const long long tok = (long long)panel * (FF_WAVES * 32) + w * 32 + l31;
    const long long tokc = tok < p.T ? tok : p.T - 1;
    const half_t* xp = p.x + tokc * p.ldx + 8 * h;
    const half_t* pe = PE ? p.rowbias + (long long)((tokc / p.rb_d1) % p.rb_md) * p.ldrb + 8 * h : nullptr;
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) {
      if (!bias_only) {
#ifdef FF_X_NOPRO      /* timing knob (tools/micro/ff_knobs.sh): no token loads.  CAUTION: constant rows make constant MFMA
                          operands, and the chip clocks higher on those - this knob overstates what the loads cost (use NOLN) */
        raw[ks] = (half8_t){1, 2, 3, 4, 5, 6, 7, (half_t)ks};
#else
        raw[ks] = *(const half8_t*)(xp + 16 * ks);
#endif
      }
      if (PE && (bias_only || !PE)) pv[ks] = *(const half8_t*)(pe + 16 * ks);
    }
  };
  // LayerNorm of x' = x (+ row bias) in registers -> fp16 MFMA operands a[160:239]; returns mean and sigma of the row
  // (No implicit contraction here: the lambda is instantiated twice - first panel, later panels - and a token row must give
  // the same bits in both, or the two CFG halves of a batch, identical rows in different panels, drift apart by an fp16 ulp.)
  auto layernorm_rows = [&](half8_t (&raw)[20], half8_t (&pv)[20], float& ln_mean, float& ln_sigma) {
#pragma clang fp contract(off)
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)raw[ks][e];
        if (PE) f += (float)pv[ks][e];
        s += f;
        q = fmaf(f, f, q);
      }
    }
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    const float mean = s * (1.0f / FF_C);
    float var = fmaf(-mean, mean, q * (1.0f / FF_C));
    var = var < 0.f ? 0.f : var;
    const float rstd = __builtin_amdgcn_rsqf(var + p.eps);
    const float nm = -mean * rstd;
    ln_mean = mean;
    ln_sigma = (var + p.eps) * rstd;
    ff_for20([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      // (opaque copies: the second pass converts the fp16 rows AGAIN instead of keeping 160 fp32 values of the first
      // pass alive - those would not fit the vector registers and the compiler would park them in accumulation registers)
      asm volatile("" : "+v"(raw[ks]));
      if (PE) asm volatile("" : "+v"(pv[ks]));
      half8_t z;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)raw[ks][e];
        if (PE) f += (float)pv[ks][e];
        z[e] = (half_t)fmaf(f, rstd, nm);
      }
      typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
      const uint4_t u = __builtin_bit_cast(uint4_t, z);
      ff_agpr_write<FF_ZF + 4 * ks + 0>(u[0]);
      ff_agpr_write<FF_ZF + 4 * ks + 1>(u[1]);
      ff_agpr_write<FF_ZF + 4 * ks + 2>(u[2]);
      ff_agpr_write<FF_ZF + 4 * ks + 3>(u[3]);
    });
  };

  float ln_mean, ln_sigma;        // the epilogue rebuilds the residual x' = z * sigma + mean from the normalised fragments
  {
    half8_t raw[20], pv[20];
    load_rows(blockIdx.x, raw, pv, false);
    if (PE) load_rows(blockIdx.x, raw, pv, true);
    layernorm_rows(raw, pv, ln_mean, ln_sigma);
  }
#pragma unroll 1
  for (int panel = blockIdx.x; panel < p.npanels; panel += gridDim.x) {
    asm volatile(FF_PANEL_ASM
                 : [splo] "+s"(splo), [sphi] "+s"(sphi)
                 : [fa0] "v"(fa0), [fa1] "v"(fa1), [fa2] "v"(fa2), [vo0] "v"(vo0), [vo1] "v"(vo1), [vo2] "v"(vo2), [vo3] "v"(vo3),
                   [vo4] "v"(vo4), [vob] "v"(vob), [hmask] "v"(hmask), [ldsw] "s"(ldsw), [lds0] "s"(lds0u), [sp0lo] "s"(sp0lo),
                   [sp0hi] "s"(sp0hi)
                 : FF_CLOBBERS);

    // ---- the NEXT panel's token rows start their way from memory now: they land under this panel's epilogue
    const int nextp = panel + (int)gridDim.x;
    half8_t nraw[20], npv[20];
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    uint4_t r2raw[20];
    if (R2) {       // the second residual of THIS panel first: loads return in order
      int lane3;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane3));
      const long long tk = (long long)panel * (FF_WAVES * 32) + w * 32 + (lane3 & 31);
      const half_t* rp = p.res2 + (tk < p.T ? tk : p.T - 1) * p.ldr2 + 8 * (lane3 >> 5);
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) r2raw[ks] = *(const uint4_t*)(rp + 16 * ks);
    }
    if (nextp < p.npanels) load_rows(nextp, nraw, npv, false);

    // ---- epilogue: lane (token, h) owns accumulator r of output tile i = channel 32 i + (r & 3) + 8 (r >> 2) + 4 h.  The
    //      residual x' is rebuilt from the normalised fragments still in a[160:239] (x' = z sigma + mean: one more fp16
    //      rounding of x' - mean, the size of x's own), which lane (token, h') holds for channels 16 ks + 8 h' + 0..7: after
    //      one half-wave exchange per dword pair - lanes 0..31 give away channels 4..7 and take the partner's 8..11 - a lane
    //      holds exactly its epilogue channels.  The same exchange on the OUTPUT pairs the two 4-channel groups of a k-step
    //      into 8 consecutive channels per lane: lanes 0..31 store channels 16 ks + 0..7, lanes 32..63 16 ks + 8..15, 16 bytes.
    int lane2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
    const int h2 = lane2 >> 5, l2 = lane2 & 31;
    const long long tok2 = (long long)panel * (FF_WAVES * 32) + w * 32 + l2;
    const long long tok2c = tok2 < p.T ? tok2 : p.T - 1;
    const bool live = tok2 < p.T;
    const float* b2p = (const float*)(smem + FF_B2_OFF) + 4 * h2;
    half_t* op = p.out + tok2c * p.ldo + 8 * h2;
    const float sa = p.s_acc, r2 = p.r2;
#ifdef FF_X_NOEPI       /* timing knob: no epilogue (one accumulator read keeps the statement alive) */
    if (live && ff_agpr_read<FF_YACC>() == 12345.678f) *op = (half_t)1.f;
    if (false)
#endif
    ff_for20([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int i = ks >> 1, g0 = 2 * (ks & 1);            // output tile, first of the k-step's two 4-channel groups
      // (a scheduling fence per k-step: without it the compiler hoists all forty bias loads, runs out of vector registers
      // and parks values in ACCUMULATION registers - the ones that still hold Y^T and z^T here)
      asm volatile("" ::: "memory");
      unsigned z[4];
      z[0] = __builtin_bit_cast(unsigned, ff_agpr_read<FF_ZF + 4 * ks + 0>());
      z[1] = __builtin_bit_cast(unsigned, ff_agpr_read<FF_ZF + 4 * ks + 1>());
      z[2] = __builtin_bit_cast(unsigned, ff_agpr_read<FF_ZF + 4 * ks + 2>());
      z[3] = __builtin_bit_cast(unsigned, ff_agpr_read<FF_ZF + 4 * ks + 3>());
      ff_swap32(z[0], z[2]);
      ff_swap32(z[1], z[3]);
      unsigned rr[4];
      if (R2) {        // the second residual: 16 bytes of the lane's STORE group, exchanged back into the accumulator layout
        const uint4_t rv = r2raw[ks];
        rr[0] = rv[0]; rr[1] = rv[1]; rr[2] = rv[2]; rr[3] = rv[3];
        ff_swap32(rr[0], rr[2]);
        ff_swap32(rr[1], rr[3]);
      }
      unsigned o[4];
      ff_static_for([&](auto jc) {
        constexpr int j = decltype(jc)::value;                  // group g0 + j: channels 32 i + 8 (g0 + j) + 4 h + 0..3
        constexpr int c0 = 32 * i + 8 * (g0 + j);
        const float4_t bb = *(const float4_t*)(b2p + c0);
        float4_t y;
        y[0] = ff_agpr_read<FF_YACC + 16 * i + 4 * (g0 + j) + 0>();
        y[1] = ff_agpr_read<FF_YACC + 16 * i + 4 * (g0 + j) + 1>();
        y[2] = ff_agpr_read<FF_YACC + 16 * i + 4 * (g0 + j) + 2>();
        y[3] = ff_agpr_read<FF_YACC + 16 * i + 4 * (g0 + j) + 3>();
        const half2_t z0 = __builtin_bit_cast(half2_t, z[2 * j]), z1 = __builtin_bit_cast(half2_t, z[2 * j + 1]);
        const float4_t xs = {fmaf((float)z0[0], ln_sigma, ln_mean), fmaf((float)z0[1], ln_sigma, ln_mean),
                             fmaf((float)z1[0], ln_sigma, ln_mean), fmaf((float)z1[1], ln_sigma, ln_mean)};
        float4_t ov = (y + bb + xs) * sa;
        if (R2) {
          const half2_t q0 = __builtin_bit_cast(half2_t, rr[2 * j]), q1 = __builtin_bit_cast(half2_t, rr[2 * j + 1]);
          ov += (float4_t){(float)q0[0], (float)q0[1], (float)q1[0], (float)q1[1]} * r2;
        }
        const half2_t a = {(half_t)ov[0], (half_t)ov[1]}, bq = {(half_t)ov[2], (half_t)ov[3]};
        o[2 * j] = __builtin_bit_cast(unsigned, a);
        o[2 * j + 1] = __builtin_bit_cast(unsigned, bq);
      }, FfIC<0>{}, FfIC<1>{});
      // lanes 0..31 (channels 0..3 | 8..11 of the k-step) give away 8..11 and take the partner's 4..7; lanes 32..63 hold
      // 8..15 afterwards: both halves have their eight channels in order
      ff_swap32(o[0], o[2]);
      ff_swap32(o[1], o[3]);
      if (live) *(uint4_t*)(op + 16 * ks) = (uint4_t){o[0], o[1], o[2], o[3]};
    });

    if (nextp < p.npanels) {
      if (PE) load_rows(nextp, nraw, npv, true);        // (the small row-bias table: L2-resident, loaded late to save registers)
#ifdef FF_X_NOLN        /* timing knob: the next panel keeps this one's operands */
      if (p.eps == 12345.f)
#endif
      layernorm_rows(nraw, npv, ln_mean, ln_sigma);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the chunks issued ahead for a panel that does not come
}

extern "C" int lkgd_ff_fused_c320(const void* x, int32_t ldx, int64_t T, const void* rowbias, int32_t ldrb, int32_t rb_d1,
                                  int32_t rb_md, const void* wstream, const float* b2, float eps, float s_acc, const void* res2,
                                  int32_t ldr2, float r2, void* out, int32_t ldo, lkgd_stream_t stream) {
  if (!x || !wstream || !b2 || !out) return LKGD_E_NULL;
  if (T <= 0 || T > 0x7fffffffLL * 64) return LKGD_E_SHAPE;
  if (ldx % 8 || ldo % 8 || ldx < FF_C || ldo < FF_C) return LKGD_E_ALIGN;
  if (!aligned16(x) || !aligned16(wstream) || !aligned16(out) || !aligned16(b2)) return LKGD_E_ALIGN;
  if (rowbias && (ldrb % 8 || ldrb < FF_C || rb_d1 <= 0 || rb_md <= 0 || !aligned16(rowbias))) return LKGD_E_SHAPE;
  if (res2 && (ldr2 % 8 || ldr2 < FF_C || !aligned16(res2))) return LKGD_E_ALIGN;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)ff_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)ff_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)ff_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)ff_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const int cus = lkgd_cu_count();      // cached per device (common.h)
  const long long npanels = (T + FF_WAVES * 32 - 1) / (FF_WAVES * 32);
  ff_params p;
  p.x = (const half_t*)x; p.ldx = ldx; p.T = T;
  p.rowbias = (const half_t*)rowbias; p.ldrb = ldrb; p.rb_d1 = rb_d1 > 0 ? rb_d1 : 1; p.rb_md = rb_md > 0 ? rb_md : 1;
  p.wstream = (const char*)wstream; p.b2 = b2; p.eps = eps; p.s_acc = s_acc; p.r2 = r2;
  p.res2 = (const half_t*)res2; p.ldr2 = ldr2; p.out = (half_t*)out; p.ldo = ldo; p.npanels = (int)npanels;
  const int grid = npanels < cus ? (int)npanels : cus;
  const dim3 gr(grid), bl(FF_WAVES * 64);
  hipStream_t st = (hipStream_t)stream;
  if (rowbias && res2) hipLaunchKernelGGL((ff_fused_kernel<true, true>), gr, bl, FF_LDS, st, p);
  else if (rowbias) hipLaunchKernelGGL((ff_fused_kernel<true, false>), gr, bl, FF_LDS, st, p);
  else if (res2) hipLaunchKernelGGL((ff_fused_kernel<false, true>), gr, bl, FF_LDS, st, p);
  else hipLaunchKernelGGL((ff_fused_kernel<false, false>), gr, bl, FF_LDS, st, p);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
