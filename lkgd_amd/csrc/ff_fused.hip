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
#define FF_LDS (FF_NSLOT * FF_SLOT)

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

#pragma unroll 1
  for (int panel = blockIdx.x; panel < p.npanels; panel += gridDim.x) {
    // ---- the wave's 32 token rows: this lane holds channels 16 ks + 8 h + 0..7 of row `tok` (the B operand layout)
    const long long tok = (long long)panel * (FF_WAVES * 32) + w * 32 + l31;
    const long long tokc = tok < p.T ? tok : p.T - 1;
    {
      const half_t* xp = p.x + tokc * p.ldx + 8 * h;
      const half_t* pe = p.rowbias ? p.rowbias + (long long)((tokc / p.rb_d1) % p.rb_md) * p.ldrb + 8 * h : nullptr;
      // two passes over the fp16 rows kept in registers (80 + 80 with a row bias): sums, then normalise + pack
      half8_t raw[20], pv[20];
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        raw[ks] = *(const half8_t*)(xp + 16 * ks);
        if (pe) pv[ks] = *(const half8_t*)(pe + 16 * ks);
      }
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)raw[ks][e];
          if (pe) f += (float)pv[ks][e];
          s += f;
          q += f * f;
        }
      }
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      const float mean = s * (1.0f / FF_C);
      float var = q * (1.0f / FF_C) - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float rstd = __builtin_amdgcn_rsqf(var + p.eps);
      const float nm = -mean * rstd;
      ff_for20([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        half8_t z;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)raw[ks][e];
          if (pe) f += (float)pv[ks][e];
          z[e] = (half_t)fmaf(f, rstd, nm);
        }
        typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
        const uint4_t u = __builtin_bit_cast(uint4_t, z);
        ff_agpr_write<FF_ZF + 4 * ks + 0>(u[0]);
        ff_agpr_write<FF_ZF + 4 * ks + 1>(u[1]);
        ff_agpr_write<FF_ZF + 4 * ks + 2>(u[2]);
        ff_agpr_write<FF_ZF + 4 * ks + 3>(u[3]);
      });
    }

    asm volatile(FF_PANEL_ASM
                 : [splo] "+s"(splo), [sphi] "+s"(sphi)
                 : [fa0] "v"(fa0), [fa1] "v"(fa1), [fa2] "v"(fa2), [vo0] "v"(vo0), [vo1] "v"(vo1), [vo2] "v"(vo2), [vo3] "v"(vo3),
                   [vo4] "v"(vo4), [vob] "v"(vob), [hmask] "v"(hmask), [ldsw] "s"(ldsw), [lds0] "s"(lds0u), [sp0lo] "s"(sp0lo),
                   [sp0hi] "s"(sp0hi)
                 : FF_CLOBBERS);

    // ---- epilogue: lane owns token row `tok`; accumulator r of output tile i is channel 32 i + (r & 3) + 8 (r >> 2) + 4 h
    int lane2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
    const int h2 = lane2 >> 5, l2 = lane2 & 31;
    const long long tok2 = (long long)panel * (FF_WAVES * 32) + w * 32 + l2;
    const long long tok2c = tok2 < p.T ? tok2 : p.T - 1;
    const bool live = tok2 < p.T;
    const half_t* xr = p.x + tok2c * p.ldx + 4 * h2;
    const half_t* per = p.rowbias ? p.rowbias + (long long)((tok2c / p.rb_d1) % p.rb_md) * p.ldrb + 4 * h2 : nullptr;
    const half_t* r2p = p.res2 ? p.res2 + tok2c * p.ldr2 + 4 * h2 : nullptr;
    const float* b2p = p.b2 + 4 * h2;
    half_t* op = p.out + tok2c * p.ldo + 4 * h2;
    const float sa = p.s_acc, r2 = p.r2;
    ff_for10([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ff_for4([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int c0 = 32 * i + 8 * g;
        const float4_t bb = *(const float4_t*)(b2p + c0);
        const half4_t xv = *(const half4_t*)(xr + c0);
        float4_t y;
        y[0] = ff_agpr_read<FF_YACC + 16 * i + 4 * g + 0>();
        y[1] = ff_agpr_read<FF_YACC + 16 * i + 4 * g + 1>();
        y[2] = ff_agpr_read<FF_YACC + 16 * i + 4 * g + 2>();
        y[3] = ff_agpr_read<FF_YACC + 16 * i + 4 * g + 3>();
        float4_t xs = {(float)xv[0], (float)xv[1], (float)xv[2], (float)xv[3]};
        if (per) {
          const half4_t pv = *(const half4_t*)(per + c0);
          xs += (float4_t){(float)pv[0], (float)pv[1], (float)pv[2], (float)pv[3]};
        }
        float4_t o = (y + bb + xs) * sa;
        if (r2p) {
          const half4_t rv = *(const half4_t*)(r2p + c0);
          o += (float4_t){(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]} * r2;
        }
        const half4_t ov = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
        if (live) *(half4_t*)(op + c0) = ov;
      });
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the chunks issued ahead for a panel that does not come
}

extern "C" int lkgd_ff_fused_c320(const void* x, int32_t ldx, int64_t T, const void* rowbias, int32_t ldrb, int32_t rb_d1,
                                  int32_t rb_md, const void* wstream, const float* b2, float eps, float s_acc, const void* res2,
                                  int32_t ldr2, float r2, void* out, int32_t ldo, lkgd_stream_t stream) {
  if (!x || !wstream || !b2 || !out) return LKGD_E_NULL;
  if (T <= 0 || T > 0x7fffffffLL * 64) return LKGD_E_SHAPE;
  if (ldx % 8 || ldo % 4 || ldx < FF_C || ldo < FF_C) return LKGD_E_ALIGN;
  if (!aligned16(x) || !aligned16(wstream) || ((uintptr_t)out & 7) || !aligned16(b2)) return LKGD_E_ALIGN;
  if (rowbias && (ldrb % 8 || ldrb < FF_C || rb_d1 <= 0 || rb_md <= 0 || !aligned16(rowbias))) return LKGD_E_SHAPE;
  if (res2 && (ldr2 % 4 || ldr2 < FF_C || ((uintptr_t)res2 & 7))) return LKGD_E_ALIGN;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)ff_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const long long npanels = (T + FF_WAVES * 32 - 1) / (FF_WAVES * 32);
  ff_params p;
  p.x = (const half_t*)x; p.ldx = ldx; p.T = T;
  p.rowbias = (const half_t*)rowbias; p.ldrb = ldrb; p.rb_d1 = rb_d1 > 0 ? rb_d1 : 1; p.rb_md = rb_md > 0 ? rb_md : 1;
  p.wstream = (const char*)wstream; p.b2 = b2; p.eps = eps; p.s_acc = s_acc; p.r2 = r2;
  p.res2 = (const half_t*)res2; p.ldr2 = ldr2; p.out = (half_t*)out; p.ldo = ldo; p.npanels = (int)npanels;
  const int grid = npanels < cus ? (int)npanels : cus;
  hipLaunchKernelGGL(ff_fused_kernel, dim3(grid), dim3(FF_WAVES * 64), FF_LDS, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
