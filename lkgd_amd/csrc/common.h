// Shared device helpers for the gfx950 kernels (wave64, MFMA f16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lkgd_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte async global -> LDS copy (global_load_lds_dwordx4): LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// exact-erf GELU of two values at once, erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below fp16 resolution).
// The GEGLU epilogue of a K=320 projection spends more VALU cycles here than the K-loop spends in the matrix pipe, so
// the form is chosen for instruction count:  gelu(x) = max(x,0) - |x| * (0.5 * poly(t) * t * exp(-x^2/2)),
// t = 1/(1 + p|x|/sqrt2)  (no sign select, no 1 +- erf), everything but rcp/exp2 on float2 (v_pk_fma_f32 / v_pk_mul_f32).
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2_t gelu_erf2(float2_t x) {
#if defined(LKGD_X_GELU_NONE)        /* timing experiment only (tools/micro/gelu_knobs.sh): what the erf arithmetic costs */
  return x;
#endif
  const float2_t ax = __builtin_elementwise_abs(x);
  const float2_t d = __builtin_elementwise_fma(ax, (float2_t)(0.3275911f * 0.70710678118654752440f), (float2_t)(1.0f));
  const float2_t z = ax * 0.84932180028801904272f;          // z^2 = (x^2 / 2) * log2(e)
  const float2_t z2 = z * z;
  float2_t t, e;
  t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  e.x = __builtin_amdgcn_exp2f(-z2.x); e.y = __builtin_amdgcn_exp2f(-z2.y);
  float2_t q = __builtin_elementwise_fma(t, (float2_t)(0.5f * 1.061405429f), (float2_t)(0.5f * -1.453152027f));
  q = __builtin_elementwise_fma(q, t, (float2_t)(0.5f * 1.421413741f));
  q = __builtin_elementwise_fma(q, t, (float2_t)(0.5f * -0.284496736f));
  q = __builtin_elementwise_fma(q, t, (float2_t)(0.5f * 0.254829592f));
  q = q * t;
  q = q * e;
  return __builtin_elementwise_fma(-q, ax, __builtin_elementwise_max(x, (float2_t)(0.0f)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// hipFuncSetAttribute and the CU count are per DEVICE.  Launchers remember, per device ordinal, that they have set their
// kernels' attributes (setting one twice is harmless, so a relaxed bit mask is all concurrent callers need).
#include <atomic>
struct lkgd_device_once {
  std::atomic<unsigned long long> mask{0};
  // returns the current device ordinal (0..63) and whether this launcher still has to initialise it; -1 on error
  int need(bool* todo) const {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return -1;
    *todo = ((mask.load(std::memory_order_acquire) >> dev) & 1ull) == 0;
    return dev;
  }
  void done(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};
// CU count of the current device, queried once per device ordinal (hipGetDeviceProperties is a slow host call; the launchers
// of the persistent kernels need the count on every launch to size their grids); 256 if the query fails
static inline int lkgd_cu_count() {
  static std::atomic<int> cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}
#define LKGD_DEVICE_ONCE_BEGIN                     \
  {                                                \
    static lkgd_device_once once_;                 \
    bool todo_ = false;                            \
    const int dev_ = once_.need(&todo_);           \
    if (dev_ < 0) return LKGD_E_LAUNCH;            \
    if (todo_) {
#define LKGD_DEVICE_ONCE_END \
      once_.done(dev_);      \
    }                        \
  }
