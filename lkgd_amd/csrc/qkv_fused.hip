// LayerNorm + to_q | to_k | to_v of the spatial transformer block at the 72x128 level (C = 320 -> 960) in ONE kernel
// (include/lkgd_hip.h section 3c; round 4): BasicTransformerBlock's `norm1 -> attn1.to_q / to_k / to_v`
// (patch/patch.py:416, :440-445).  The rows are read once, the normalised copy is never written, and the projection runs on the
// skeleton of ff_fused.hip / attn_tblock.hip: a wave owns 32 token rows as MFMA operands (LayerNorm-ed in registers, parked in
// a[0:79]), a workgroup = 4 waves = a 128-token panel, the 960 x 320 weights stream L2 -> LDS once per panel as 30 chunks in the
// order the generated statement consumes them (tools/gen_qkv_asm.py -> qkv_fused_loop.inc, packing.pack_ln_proj).  The statement
// also STORES: tile n's accumulators leave as two 16-byte pieces per lane (after a half-wave exchange) in the MFMA gaps of tile n + 1, and it fetches the next
// panel's token rows into its output registers - this file is the first panel's load, the LayerNorm and the launch.
#include "common.h"
#include <utility>
#include "qkv_fused_loop.inc"
#include "qkv640_fused_loop.inc"        // QKV_GEN_C=640: the 36x64 level (60 tiles, two chunks per tile, z in a[0:159])

#define QK_WAVES 4
#define QK_LDS (QK_NSLOT * QK_SLOT)

template <int REG>
__device__ __forceinline__ void qk_agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v), "i"(REG));
}
template <int V> struct QkIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void qk_static_for(F&& f, QkIC<Is>...) { (f(QkIC<Is>{}), ...); }
template <class F, int... Is> __device__ __forceinline__ void qk_for_seq(F&& f, std::integer_sequence<int, Is...>) { (f(QkIC<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void qk_for(F&& f) { qk_for_seq(f, std::make_integer_sequence<int, N>{}); }

struct qk_params {
  const half_t* x; int ldx; long long T;
  const char* wstream;
  float eps;
  half_t* out; int ldo;
  int npanels;
};

template <int C>
__global__ __launch_bounds__(QK_WAVES * 64, 1) __attribute__((amdgpu_num_vgpr(256))) void ln_qkv_kernel(qk_params p) {
  constexpr int NKS = C / 16, PARTS = NKS / 20;
  static_assert(QK_SLOT == QK6_SLOT && QK_NSLOT == QK6_NSLOT && QK_AHEAD == QK6_AHEAD && QK_ZF == 0 && QK6_ZF == 0, "one ring layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  if ((int)blockIdx.x >= p.npanels) return;

  // ---- chunks 0..AHEAD-1 of the stream (every panel's statement issues the chunks AHEAD ahead, across panel borders); the chunk
  //      that opens a tile carries the bias fragment as a 21st KiB
  unsigned long long ahead_bytes = 0;
#pragma unroll
  for (int c = 0; c < QK_AHEAD; ++c) {
    const char* src = p.wstream + ahead_bytes + lane * 16;
    char* dst = smem + c * QK_SLOT;
#pragma unroll
    for (int j = 0; j < 5; ++j) glds16(src + (w + 4 * j) * 1024, dst + (w + 4 * j) * 1024);
    if (c % PARTS == 0) glds16(src + 20480, dst + 20480);
    ahead_bytes += c % PARTS == 0 ? QK_W1_BYTES : QK_W2_BYTES;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned long long sp0 = (unsigned long long)(uintptr_t)p.wstream;
  const unsigned sp0lo = __builtin_amdgcn_readfirstlane((unsigned)sp0), sp0hi = __builtin_amdgcn_readfirstlane((unsigned)(sp0 >> 32));
  unsigned splo, sphi;
  {
    const unsigned long long sp = sp0 + ahead_bytes;
    splo = __builtin_amdgcn_readfirstlane((unsigned)sp);
    sphi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
  }
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)w * 1024u);
  const unsigned lds0u = __builtin_amdgcn_readfirstlane(lds0);
  const unsigned fa0 = lds0 + lane * 16, fa1 = fa0 + 2 * QK_SLOT, fa2 = fa0 + 4 * QK_SLOT;
  const unsigned vo0 = (unsigned)(w * 1024 + lane * 16), vo1 = vo0 + 4096, vo2 = vo0 + 8192, vo3 = vo0 + 12288, vo4 = vo0 + 16384;
  const unsigned vob = 20480u + lane * 16;
  const unsigned hmask = h == 0 ? 0xffffffffu : 0u;

  // the lane's token of a panel (rows beyond T: the last row again - its results are the last row's, stored to the same place)
  auto tok_of = [&](int panel) -> long long {
    const long long tok = (long long)panel * (QK_WAVES * 32) + w * 32 + l31;
    return tok < p.T ? tok : p.T - 1;
  };
  // LayerNorm in registers -> fp16 MFMA operands a[0:79].  (No implicit contraction: the lambda is instantiated twice and a row
  // must give the same bits in both - see ff_fused.hip.)
  auto layernorm_rows = [&](half8_t (&raw)[NKS]) {
#pragma clang fp contract(off)
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)raw[ks][e];
        s += f;
        q = fmaf(f, f, q);
      }
    }
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    const float mean = s * (1.0f / C);
    float var = fmaf(-mean, mean, q * (1.0f / C));
    var = var < 0.f ? 0.f : var;
    const float rstd = __builtin_amdgcn_rsqf(var + p.eps);
    const float nm = -mean * rstd;
    qk_for<NKS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      asm volatile("" : "+v"(raw[ks]));
      half8_t z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (half_t)fmaf((float)raw[ks][e], rstd, nm);
      typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
      const uint4_t u = __builtin_bit_cast(uint4_t, z);
      qk_agpr_write<QK_ZF + 4 * ks + 0>(u[0]);
      qk_agpr_write<QK_ZF + 4 * ks + 1>(u[1]);
      qk_agpr_write<QK_ZF + 4 * ks + 2>(u[2]);
      qk_agpr_write<QK_ZF + 4 * ks + 3>(u[3]);
    });
  };

  auto load_rows = [&](int panel, half8_t (&raw)[NKS]) {
    const half_t* xp = p.x + tok_of(panel) * p.ldx + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) raw[ks] = *(const half8_t*)(xp + 16 * ks);
  };
  {
    half8_t raw[NKS];
    load_rows(blockIdx.x, raw);
    layernorm_rows(raw);
  }
  // (every statement starts with a COUNTED wait that leaves its predecessor's last stores in flight: the first one has no
  // predecessor, so everything issued so far - the look-ahead chunks - is waited for here)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int panel = blockIdx.x; panel < p.npanels; panel += gridDim.x) {
    const int nextp = panel + (int)gridDim.x;
    half_t* orow = p.out + tok_of(panel) * p.ldo + 8 * h;          // (the upper half-wave stores the second 16 bytes of a pair)
    splo = __builtin_amdgcn_readfirstlane(splo);
    sphi = __builtin_amdgcn_readfirstlane(sphi);
    if constexpr (C == 320) {
      // the statement stores this panel's 960 channels per row and fetches the NEXT panel's rows (a workgroup's last panel
      // fetches its own again and drops them)
      const half_t* xrow = p.x + tok_of(nextp < p.npanels ? nextp : panel) * p.ldx + 8 * h;
      half8_t nraw[NKS];
#define QK_ROW_OUT(i) [r##i] "=&v"(nraw[i])
      asm volatile(QK_PANEL_ASM
                   : [splo] "+s"(splo), [sphi] "+s"(sphi), QK_ROW_OUT(0), QK_ROW_OUT(1), QK_ROW_OUT(2), QK_ROW_OUT(3), QK_ROW_OUT(4),
                     QK_ROW_OUT(5), QK_ROW_OUT(6), QK_ROW_OUT(7), QK_ROW_OUT(8), QK_ROW_OUT(9), QK_ROW_OUT(10), QK_ROW_OUT(11),
                     QK_ROW_OUT(12), QK_ROW_OUT(13), QK_ROW_OUT(14), QK_ROW_OUT(15), QK_ROW_OUT(16), QK_ROW_OUT(17), QK_ROW_OUT(18),
                     QK_ROW_OUT(19)
                   : [xrow] "v"(xrow), [orow] "v"(orow), [fa0] "v"(fa0), [fa1] "v"(fa1), [fa2] "v"(fa2), [vo0] "v"(vo0), [vo1] "v"(vo1),
                     [vo2] "v"(vo2), [vo3] "v"(vo3), [vo4] "v"(vo4), [vob] "v"(vob), [hmask] "v"(hmask), [ldsw] "s"(ldsw),
                     [lds0] "s"(lds0u), [sp0lo] "s"(sp0lo), [sp0hi] "s"(sp0hi)
                   : QK_CLOBBERS);
      if (nextp < p.npanels) layernorm_rows(nraw);
    } else {
      // 640 channels: 160 row registers do not fit beside the statement's - the next panel's rows are loaded behind it
      asm volatile(QK6_PANEL_ASM
                   : [splo] "+s"(splo), [sphi] "+s"(sphi)
                   : [orow] "v"(orow), [fa0] "v"(fa0), [fa1] "v"(fa1), [fa2] "v"(fa2), [vo0] "v"(vo0), [vo1] "v"(vo1), [vo2] "v"(vo2),
                     [vo3] "v"(vo3), [vo4] "v"(vo4), [vob] "v"(vob), [hmask] "v"(hmask), [ldsw] "s"(ldsw), [lds0] "s"(lds0u),
                     [sp0lo] "s"(sp0lo), [sp0hi] "s"(sp0hi)
                   : QK6_CLOBBERS);
      half8_t nraw[NKS];
      load_rows(nextp < p.npanels ? nextp : panel, nraw);
      if (nextp < p.npanels) layernorm_rows(nraw);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last stores; the chunks issued ahead for a panel that does not come
}

template <int C>
static int ln_qkv_launch(const void* x, int32_t ldx, int64_t T, const void* wstream, float eps, void* out, int32_t ldo,
                         lkgd_stream_t stream) {
  if (!x || !wstream || !out) return LKGD_E_NULL;
  if (T <= 0 || T > 0x7fffffffLL * 64) return LKGD_E_SHAPE;
  if (ldx % 8 || ldo % 8 || ldx < C || ldo < 3 * C) return LKGD_E_ALIGN;
  if (!aligned16(x) || !aligned16(wstream) || !aligned16(out)) return LKGD_E_ALIGN;
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)ln_qkv_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, QK_LDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  const int cus = lkgd_cu_count();      // cached per device (common.h)
  const long long npanels = (T + QK_WAVES * 32 - 1) / (QK_WAVES * 32);
  qk_params p;
  p.x = (const half_t*)x; p.ldx = ldx; p.T = T; p.wstream = (const char*)wstream; p.eps = eps;
  p.out = (half_t*)out; p.ldo = ldo; p.npanels = (int)npanels;
  const int grid = npanels < cus ? (int)npanels : cus;
  hipLaunchKernelGGL(ln_qkv_kernel<C>, dim3(grid), dim3(QK_WAVES * 64), QK_LDS, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_ln_qkv_c320(const void* x, int32_t ldx, int64_t T, const void* wstream, float eps, void* out, int32_t ldo,
                                lkgd_stream_t stream) {
  return ln_qkv_launch<320>(x, ldx, T, wstream, eps, out, ldo, stream);
}
extern "C" int lkgd_ln_qkv_c640(const void* x, int32_t ldx, int64_t T, const void* wstream, float eps, void* out, int32_t ldo,
                                lkgd_stream_t stream) {
  return ln_qkv_launch<640>(x, ldx, T, wstream, eps, out, ldo, stream);
}
