// GroupNorm(32)+SiLU and LayerNorm on channels-last fp16 token matrices (include/lkgd_hip.h sections 2, 3).
// HBM-bound kernels: 16-byte loads/stores, fp32 statistics, deterministic reductions (no float atomics to HBM).
#include "common.h"

// rows of one sample handled by one workgroup.  Large maps: ~32 KiB of activations per apply workgroup, 128 KiB per
// statistics workgroup (fewer, larger partial sums; measured best at the 72x128 level).  Small maps - the 18x32 / 9x16
// levels, every level of a frame-sharded rank - are cut finer, down to 8 KiB, so that a launch still has about GN_TARGET_WGS
// workgroups: with 128-KiB chunks the 18x32 map of a rank of 8 (5.9 MB) was 48 workgroups walking 512 bytes per thread one
// 8-load batch after the other (8-10 us per launch, profiles/r05_plan_profile_base.txt).
static thread_local int gn_apply_kb = 32;
extern "C" void lkgd_debug_set_gn_apply_kb(int kb) { gn_apply_kb = kb < 32 ? 32 : kb; }
static thread_local int gn_stats_kb = 128;
extern "C" void lkgd_debug_set_gn_stats_kb(int kb) { gn_stats_kb = kb < 32 ? 32 : kb; }
static thread_local int gn_target_wgs = 1024;
extern "C" void lkgd_debug_set_gn_target_wgs(int n) { gn_target_wgs = n < 1 ? 1 : n; }   // 1 = the fixed chunk sizes only
static inline int gn_rows_kb(int C, int kb, long long rows_per_sample, long long nsamples) {
  int rmax = kb * 1024 / (C * 2);
  rmax = rmax < 8 ? 8 : (rmax > 1024 ? 1024 : rmax);
  int rmin = 8 * 1024 / (C * 2);
  rmin = rmin < 8 ? 8 : rmin;
  if (rmin > rmax) rmin = rmax;
  const long long want = (rows_per_sample * nsamples + gn_target_wgs - 1) / gn_target_wgs;   // rows per chunk for ~target chunks
  return (int)(want < rmin ? rmin : (want > rmax ? rmax : want));
}
static inline int gn_rows(int C, long long rows_per_sample, long long nsamples) {
  return gn_rows_kb(C, gn_apply_kb, rows_per_sample, nsamples);
}
// the statistics pass may use longer chunks than the apply pass (the partial buffer is sized by lkgd_groupnorm_chunks)
static inline int gn_rows_stats(int C, long long rows_per_sample, long long nsamples) {
  return gn_rows_kb(C, gn_stats_kb, rows_per_sample, nsamples);
}
#define GN_GROUPS 32
#define GN_MAXC 4096

// thread -> (row lane rp, 16-byte column vector cv); a thread's channel set is fixed for the whole kernel
struct GnMap {
  int C8, rows_par, nslot;
};
__device__ __forceinline__ GnMap gn_map(int C) {
  GnMap m;
  m.C8 = C >> 3;
  if (m.C8 <= 256) { m.rows_par = 256 / m.C8; m.nslot = 1; }
  else { m.rows_par = 1; m.nslot = (m.C8 + 255) / 256; }
  return m;
}

__device__ __forceinline__ half8_t gn_load(const half_t* x0, int c0, int ld0, const half_t* x1, int ld1,
                                            long long row, int cv) {
  int c = cv << 3;
  if (c < c0) return *(const half8_t*)(x0 + row * ld0 + c);
  return *(const half8_t*)(x1 + row * ld1 + (c - c0));
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* x0, int c0, int ld0, const half_t* x1, int c1,
                                                       int ld1, long long rows_per_sample, float* partial,
                                                       int nchunks, int GN_ROWS) {
  __shared__ float s_sum[GN_MAXC];
  __shared__ float s_sq[GN_MAXC];
  const int C = c0 + c1;
  const GnMap mp = gn_map(C);
  const int t = threadIdx.x;
  const int chunk = blockIdx.x;
  const long long sample = blockIdx.y;
  // s_sum/s_sq hold one [C] row per row-lane rp (rows_par * C <= 2048 floats, or C <= 4096 when rows_par == 1)
  const long long r0 = (long long)chunk * GN_ROWS;
  long long r1 = r0 + GN_ROWS;
  if (r1 > rows_per_sample) r1 = rows_per_sample;
  const long long base = sample * rows_per_sample;
  for (int slot = 0; slot < mp.nslot; ++slot) {
    int cv, rp;
    if (mp.nslot == 1) { rp = t / mp.C8; cv = t - rp * mp.C8; if (rp >= mp.rows_par) continue; }
    else { rp = 0; cv = t + 256 * slot; if (cv >= mp.C8) continue; }
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll 8
    for (long long r = r0 + rp; r < r1; r += mp.rows_par) {      // 8 x 16 B in flight per thread
      half8_t v = gn_load(x0, c0, ld0, x1, ld1, base + r, cv);
#pragma unroll
      for (int e = 0; e < 8; ++e) { float f = (float)v[e]; s[e] += f; q[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s_sum[rp * C + cv * 8 + e] = s[e];
      s_sq[rp * C + cv * 8 + e] = q[e];
    }
  }
  __syncthreads();
  if (t < GN_GROUPS) {   // fixed-order reduction: deterministic
    const int gs = C / GN_GROUPS;
    float a = 0.f, b = 0.f;
    for (int rp = 0; rp < mp.rows_par; ++rp)
      for (int c = t * gs; c < (t + 1) * gs; ++c) { a += s_sum[rp * C + c]; b += s_sq[rp * C + c]; }
    float* o = partial + ((sample * nchunks + chunk) * GN_GROUPS + t) * 2;
    o[0] = a; o[1] = b;
  }
}

// Both reduction stages run in a fixed order: results are bitwise reproducible.  Statistics are fp32 over fp16 data,
// combined across chunks in fp64.  One wave per (sample, group): lanes stride over the chunks, fixed-order butterfly.
__device__ __forceinline__ void gn_reduce_pair(const float* partial, int nchunks, long long sample, int g, int lane,
                                               double& sa, double& sb) {
  double a = 0.0, b = 0.0;
  for (int c = lane; c < nchunks; c += 64) {
    const float* p = partial + ((sample * nchunks + c) * GN_GROUPS + g) * 2;
    a += p[0]; b += p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  sa = a; sb = b;
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* partial, int nchunks, double inv_count,
                                                          float eps, float* stats) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);      // 8 workgroups x 4 waves = 32 groups
  const long long sample = blockIdx.y;
  double sa, sb;
  gn_reduce_pair(partial, nchunks, sample, g, lane, sa, sb);
  if (lane == 0) {
    double mean = sa * inv_count;
    double var = sb * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(sample * GN_GROUPS + g) * 2 + 0] = (float)mean;
    stats[(sample * GN_GROUPS + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Statistics from the column sums the producing GEMMs left (lkgd_gemm_desc.colstats): one workgroup per (sample, group)
// walks the group's channels x the sample's row blocks in a fixed thread -> element map and reduces in a fixed tree
// (bitwise reproducible; fp64 across blocks).  A group may straddle the two sources of a concatenated input.
__global__ __launch_bounds__(256) void gn_cols_kernel(const float* cs0, int blk0, int ld0, int c0, const float* cs1, int blk1,
                                                      int ld1, int c1, long long rows_per_sample, double inv_count, float eps,
                                                      int as_sums, float* stats) {
  __shared__ double sa[256], sb[256];
  const int t = threadIdx.x;
  const int g = blockIdx.x;
  const long long sample = blockIdx.y;
  const int gs = (c0 + c1) / GN_GROUPS;
  double a = 0.0, b = 0.0;
  for (int src = 0; src < 2; ++src) {
    const float* cs = src ? cs1 : cs0;
    if (!cs) continue;
    const int blk = src ? blk1 : blk0, ld = src ? ld1 : ld0;
    // this group's channels inside the source: [lo, hi) in the source's own channel numbering
    int lo = g * gs - (src ? c0 : 0), hi = lo + gs;
    const int cmax = src ? c1 : c0;
    lo = lo < 0 ? 0 : lo;
    hi = hi > cmax ? cmax : hi;
    const int nch = hi - lo;
    if (nch <= 0) continue;
    // the column sums are kept per channel PAIR (2c, 2c + 1): groups and sources are even-sized
    const int plo = lo >> 1, npair = nch >> 1;
    const long long nb = rows_per_sample / blk, b0 = sample * nb;
    const long long total = nb * npair;
    for (long long e = t; e < total; e += 256) {
      const long long bi = e / npair;
      const int c = plo + (int)(e - bi * npair);
      const float* q = cs + ((b0 + bi) * (ld >> 1) + c) * 2;
      a += q[0]; b += q[1];
    }
  }
  sa[t] = a; sb[t] = b;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { sa[t] += sa[t + o]; sb[t] += sb[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    float* o = stats + (sample * GN_GROUPS + g) * 2;
    if (as_sums) {
      o[0] = (float)sa[0]; o[1] = (float)sb[0];
    } else {
      const double mean = sa[0] * inv_count;
      double var = sb[0] * inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      o[0] = (float)mean;
      o[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
}

// frame-sharded path: chunk partials -> raw fp32 sums per (sample, group); finalise from (all-reduced) sums
__global__ __launch_bounds__(256) void gn_sums_kernel(const float* partial, int nchunks, float* sums) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long sample = blockIdx.y;
  double sa, sb;
  gn_reduce_pair(partial, nchunks, sample, g, lane, sa, sb);
  if (lane == 0) {
    sums[(sample * GN_GROUPS + g) * 2 + 0] = (float)sa;
    sums[(sample * GN_GROUPS + g) * 2 + 1] = (float)sb;
  }
}
__global__ void gn_finalize_sums_kernel(const float* sums, long long n, double inv_count, float eps, float* stats) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double mean = (double)sums[i * 2] * inv_count;
  double var = (double)sums[i * 2 + 1] * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// (round 5 measured the finalize step inside this kernel's prologue - every workgroup reducing the chunk partials of its
// sample, bitwise the finalize kernel's numbers, one launch less per GroupNorm - on the maps of a rank of 8 and of the full
// forward: 22.2 vs 21.6 ms and 89.7 vs 88.9 ms per forward, SLOWER both times, as round 3 had found at full size;
// profiles/r05_gn_ab.txt.  Removed again.)
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* x0, int c0, int ld0, const half_t* x1, int c1,
                                                       int ld1, long long rows_per_sample, const float* stats,
                                                       const float* gamma, const float* beta, int silu,
                                                       half_t* out, int ldo, int GN_ROWS) {
  const int C = c0 + c1;
  const GnMap mp = gn_map(C);
  const int t = threadIdx.x;
  const long long sample = blockIdx.y;
  const long long r0 = (long long)blockIdx.x * GN_ROWS;
  long long r1 = r0 + GN_ROWS;
  if (r1 > rows_per_sample) r1 = rows_per_sample;
  const long long base = sample * rows_per_sample;
  const int gs = C / GN_GROUPS;
  for (int slot = 0; slot < mp.nslot; ++slot) {
    int cv, rp;
    if (mp.nslot == 1) { rp = t / mp.C8; cv = t - rp * mp.C8; if (rp >= mp.rows_par) continue; }
    else { rp = 0; cv = t + 256 * slot; if (cv >= mp.C8) continue; }
    // per-channel scale/shift of this thread's 8 channels: 16-byte loads of gamma/beta, 8-byte loads of (mean, rstd)
    float A[8], B[8];
    {
      const float4_t g0 = *(const float4_t*)(gamma + cv * 8), g1 = *(const float4_t*)(gamma + cv * 8 + 4);
      const float4_t b0 = *(const float4_t*)(beta + cv * 8), b1 = *(const float4_t*)(beta + cv * 8 + 4);
      typedef float float2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (cv * 8 + e) / gs;
        const float2_t mr = *(const float2_t*)(stats + (sample * GN_GROUPS + g) * 2);
        const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
        A[e] = mr[1] * ga;
        B[e] = be - mr[0] * A[e];
      }
    }
#pragma unroll 4
    for (long long r = r0 + rp; r < r1; r += mp.rows_par) {
      half8_t v = gn_load(x0, c0, ld0, x1, ld1, base + r, cv);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[e] * A[e] + B[e];
        if (silu) f = silu_f(f);
        o[e] = (half_t)f;
      }
      *(half8_t*)(out + (base + r) * ldo + cv * 8) = o;
    }
  }
}

static int gn_check(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, long long nsamples,
                    long long rows) {
  if (!x0) return LKGD_E_NULL;
  if (c1 > 0 && !x1) return LKGD_E_NULL;
  int C = c0 + c1;
  if (c0 <= 0 || c1 < 0 || C % GN_GROUPS || C % 8 || c0 % 8 || C > GN_MAXC) return LKGD_E_SHAPE;
  if (ld0 % 8 || (c1 > 0 && ld1 % 8)) return LKGD_E_ALIGN;
  if (!aligned16(x0) || (x1 && !aligned16(x1))) return LKGD_E_ALIGN;
  if (nsamples <= 0 || rows <= 0 || nsamples > 65535) return LKGD_E_SHAPE;
  return LKGD_OK;
}

extern "C" int lkgd_groupnorm_chunks(int64_t rows_per_sample, int32_t C) {   // sizes the caller's `partial` scratch
  // chunk rows never shrink with the sample count: one sample gives the largest chunk count
  const int ra = gn_rows(C, rows_per_sample, 1), rs = gn_rows_stats(C, rows_per_sample, 1);
  const int r = ra < rs ? ra : rs;
  return (int)((rows_per_sample + r - 1) / r);
}

extern "C" int lkgd_groupnorm_stats(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                                    int64_t nsamples, int64_t rows_per_sample, float eps, float* partial,
                                    float* stats, lkgd_stream_t stream) {
  int rc = gn_check(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample);
  if (rc) return rc;
  if (!partial || !stats) return LKGD_E_NULL;
  const int rs = gn_rows_stats(c0 + c1, rows_per_sample, nsamples);
  int nchunks = (int)((rows_per_sample + rs - 1) / rs);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x0, c0, ld0, (const half_t*)x1, c1, ld1, (long long)rows_per_sample, partial,
                     nchunks, rs);
  double inv = 1.0 / ((double)rows_per_sample * (double)((c0 + c1) / GN_GROUPS));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(8, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream, partial,
                     nchunks, inv, eps, stats);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_groupnorm_stats_cols(const float* cs0, int32_t blk0, int32_t ldcs0, int32_t c0, const float* cs1,
                                         int32_t blk1, int32_t ldcs1, int32_t c1, int64_t nsamples, int64_t rows_per_sample,
                                         float eps, int32_t as_sums, float* stats, lkgd_stream_t stream) {
  if (!cs0 || !stats || (c1 > 0 && !cs1)) return LKGD_E_NULL;
  const int C = c0 + c1;
  if ((C / GN_GROUPS) % 2 || c0 % 2 || c1 % 2 || ldcs0 % 2 || ldcs1 % 2) return LKGD_E_SHAPE;      // pair granularity
  if (c0 <= 0 || c1 < 0 || C % GN_GROUPS || blk0 <= 0 || ldcs0 < c0 || (c1 > 0 && (blk1 <= 0 || ldcs1 < c1))) return LKGD_E_SHAPE;
  if (nsamples <= 0 || nsamples > 65535 || rows_per_sample <= 0 || rows_per_sample % blk0 || (c1 > 0 && rows_per_sample % blk1))
    return LKGD_E_SHAPE;
  const double inv = 1.0 / ((double)rows_per_sample * (double)(C / GN_GROUPS));
  hipLaunchKernelGGL(gn_cols_kernel, dim3(GN_GROUPS, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream, cs0, blk0, ldcs0,
                     c0, c1 > 0 ? cs1 : (const float*)nullptr, blk1, ldcs1, c1, (long long)rows_per_sample, inv, eps, as_sums,
                     stats);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_groupnorm_sums(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                                   int64_t nsamples, int64_t rows_per_sample, float* partial, float* sums,
                                   lkgd_stream_t stream) {
  int rc = gn_check(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample);
  if (rc) return rc;
  if (!partial || !sums) return LKGD_E_NULL;
  const int rs = gn_rows_stats(c0 + c1, rows_per_sample, nsamples);
  int nchunks = (int)((rows_per_sample + rs - 1) / rs);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x0, c0, ld0, (const half_t*)x1, c1, ld1, (long long)rows_per_sample, partial,
                     nchunks, rs);
  hipLaunchKernelGGL(gn_sums_kernel, dim3(8, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream, partial, nchunks,
                     sums);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// the same finalize over the partial sums of several ranks, added in rank order in fp64 (every rank of a frame group runs it on
// the same gathered buffer: identical statistics everywhere, no all-reduce): part r's sums of sample s start at
// parts + r * part_stride + s * sample_stride (floats)
__global__ void gn_finalize_parts_kernel(const float* parts, int nparts, long long part_stride, long long sample_stride,
                                         long long n, double inv_count, float eps, float* stats) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long sample = i / GN_GROUPS;
  const int g = (int)(i - sample * GN_GROUPS);
  double a = 0.0, b = 0.0;
  for (int r = 0; r < nparts; ++r) {
    const float* p = parts + r * part_stride + sample * sample_stride + g * 2;
    a += (double)p[0]; b += (double)p[1];
  }
  double mean = a * inv_count;
  double var = b * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" int lkgd_groupnorm_finalize_parts(const float* parts, int32_t nparts, int64_t part_stride, int64_t nsamples,
                                             int64_t sample_stride, double count_per_group, float eps, float* stats,
                                             lkgd_stream_t stream) {
  if (!parts || !stats) return LKGD_E_NULL;
  if (nparts <= 0 || nsamples <= 0 || part_stride < 0 || sample_stride < GN_GROUPS * 2 || !(count_per_group > 0.0)) return LKGD_E_SHAPE;
  if (((uintptr_t)parts & 3u) != 0) return LKGD_E_ALIGN;
  long long n = nsamples * GN_GROUPS;
  hipLaunchKernelGGL(gn_finalize_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts,
                     nparts, (long long)part_stride, (long long)sample_stride, n, 1.0 / count_per_group, eps, stats);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_groupnorm_finalize(const float* sums, int64_t nsamples, double count_per_group, float eps,
                                       float* stats, lkgd_stream_t stream) {
  if (!sums || !stats) return LKGD_E_NULL;
  if (nsamples <= 0 || !(count_per_group > 0.0)) return LKGD_E_SHAPE;
  long long n = nsamples * GN_GROUPS;
  hipLaunchKernelGGL(gn_finalize_sums_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     sums, n, 1.0 / count_per_group, eps, stats);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

extern "C" int lkgd_groupnorm_apply(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                                    int64_t nsamples, int64_t rows_per_sample, const float* stats,
                                    const float* gamma, const float* beta, int32_t silu, void* out, int32_t ldo,
                                    lkgd_stream_t stream) {
  int rc = gn_check(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample);
  if (rc) return rc;
  if (!stats || !gamma || !beta || !out) return LKGD_E_NULL;
  if (ldo % 8 || !aligned16(out)) return LKGD_E_ALIGN;
  const int ra = gn_rows(c0 + c1, rows_per_sample, nsamples);
  int nchunks = (int)((rows_per_sample + ra - 1) / ra);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunks, (unsigned)nsamples), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x0, c0, ld0, (const half_t*)x1, c1, ld1, (long long)rows_per_sample, stats, gamma,
                     beta, silu, (half_t*)out, ldo, ra);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// GroupNorm apply over a TABLE of row segments (frame-sharded ranks): segment s = rows [0, seg[s].rows) of a [rows, C] matrix at
// seg[s].src, normalised with the statistics of sample seg[s].sample, written to seg[s].dst.  One launch normalises a rank's own
// frames of every batch entry into the middle of its [F + 2]-frame buffers AND the raw boundary frames it received from its two
// neighbours into the end slots (lkgd_amd/unet.py::_temporal_norm), instead of three launches per entry.
struct gn_segment { const half_t* src; half_t* dst; long long rows; long long sample; };
__global__ __launch_bounds__(256) void gn_apply_segments_kernel(const gn_segment* segs, int C, int ld, const float* stats,
                                                                const float* gamma, const float* beta, int silu, int GN_ROWS) {
  const gn_segment sg = segs[blockIdx.y];
  const long long r0 = (long long)blockIdx.x * GN_ROWS;
  if (r0 >= sg.rows) return;
  long long r1 = r0 + GN_ROWS;
  if (r1 > sg.rows) r1 = sg.rows;
  const GnMap mp = gn_map(C);
  const int t = threadIdx.x;
  const int gs = C / GN_GROUPS;
  for (int slot = 0; slot < mp.nslot; ++slot) {
    int cv, rp;
    if (mp.nslot == 1) { rp = t / mp.C8; cv = t - rp * mp.C8; if (rp >= mp.rows_par) continue; }
    else { rp = 0; cv = t + 256 * slot; if (cv >= mp.C8) continue; }
    float A[8], B[8];
    {
      const float4_t g0 = *(const float4_t*)(gamma + cv * 8), g1 = *(const float4_t*)(gamma + cv * 8 + 4);
      const float4_t b0 = *(const float4_t*)(beta + cv * 8), b1 = *(const float4_t*)(beta + cv * 8 + 4);
      typedef float float2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (cv * 8 + e) / gs;
        const float2_t mr = *(const float2_t*)(stats + (sg.sample * GN_GROUPS + g) * 2);
        const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
        A[e] = mr[1] * ga;
        B[e] = be - mr[0] * A[e];
      }
    }
#pragma unroll 4
    for (long long r = r0 + rp; r < r1; r += mp.rows_par) {
      const half8_t v = *(const half8_t*)(sg.src + r * ld + cv * 8);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[e] * A[e] + B[e];       // the arithmetic of gn_apply_kernel: a row normalised here or there has the same bits
        if (silu) f = silu_f(f);
        o[e] = (half_t)f;
      }
      *(half8_t*)(sg.dst + r * ld + cv * 8) = o;
    }
  }
}

extern "C" int lkgd_groupnorm_apply_segments(const void* segs, int32_t nseg, int64_t max_rows, int32_t C, int32_t ld,
                                             const float* stats, const float* gamma, const float* beta, int32_t silu,
                                             lkgd_stream_t stream) {
  if (!segs || !stats || !gamma || !beta) return LKGD_E_NULL;
  if (nseg <= 0 || nseg > 65535 || max_rows <= 0 || C <= 0 || C % GN_GROUPS || C % 8 || C > GN_MAXC || ld % 8 || ld < C) return LKGD_E_SHAPE;
  if (((uintptr_t)segs & 7u) != 0) return LKGD_E_ALIGN;
  const int ra = gn_rows(C, max_rows, nseg);
  const int nchunks = (int)((max_rows + ra - 1) / ra);
  hipLaunchKernelGGL(gn_apply_segments_kernel, dim3(nchunks, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream,
                     (const gn_segment*)segs, C, ld, stats, gamma, beta, silu, ra);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}

// ---- small samples: the WHOLE GroupNorm (+ SiLU) in one launch.  One workgroup per (sample, group) holds its group's values
// (rows x C/32 channels) in LDS: read once (fp16 -> LDS, fp32 sums on the way), reduce, normalise out of LDS, write.  For the
// 18x32 / 9x16 levels (46 / 12 KiB per group) and the 36x64 level of the spatial norms (92 KiB): three dependent launches of
// 5-8 us each become one - what counts on a sharded rank, whose GroupNorms are launch-bound (22.9 us per call on a rank of
// 8 for 0.57 ms of bytes; profiles/r05_plan_profile_forms.txt).  VEC = halfs per access (C/32 = 40, 80 -> 8; 20, 60 -> 4;
// 10, 30 -> 2): the group's channels of a row are contiguous, the thread -> element map and the reduction tree are fixed
// (bitwise reproducible), the variance is E[x^2] - mean^2 in fp64 like gn_finalize_kernel's.
#define GN_SMALL_NT 512
#define GN_SMALL_U 4           // loads in flight per thread (one per iteration leaves a workgroup at 4 GB/s: 48 us for 90 KiB)
template <int VEC>
__global__ __launch_bounds__(GN_SMALL_NT) void gn_small_kernel(const half_t* x0, int c0, int ld0, const half_t* x1, int c1, int ld1,
                                                               int rows, float eps, const float* gamma, const float* beta, int silu,
                                                               half_t* out, int ldo, float* stats) {
  typedef half_t hv_t __attribute__((ext_vector_type(VEC)));
  extern __shared__ __attribute__((aligned(16))) char gn_smem[];
  hv_t* buf = (hv_t*)gn_smem;
  __shared__ double red[2 * (GN_SMALL_NT / 64)];
  __shared__ float mr[2];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int g = blockIdx.x;
  const long long sample = blockIdx.y;
  const int gs = (c0 + c1) / GN_GROUPS, per = gs / VEC;           // channels per group, accesses per row
  const int units = rows * per;
  const long long base = sample * rows;
  // element u = (row u / per, access u % per): stepped without divisions (NT = q * per + rem)
  const int q = GN_SMALL_NT / per, rem = GN_SMALL_NT - q * per;
  int r = t / per, a = t - r * per;
  float s = 0.f, ss = 0.f;
  for (int u = t; u < units; u += GN_SMALL_NT * GN_SMALL_U) {
    hv_t v[GN_SMALL_U];
#pragma unroll
    for (int k = 0; k < GN_SMALL_U; ++k) {
      const int c = g * gs + a * VEC;
      const bool ok = u + k * GN_SMALL_NT < units;
      const long long row = ok ? base + r : base;                  // (past the end: re-read row 0, never used)
      const half_t* src = c < c0 ? x0 + row * ld0 + c : x1 + row * ld1 + (c - c0);
      v[k] = *(const hv_t*)src;
      r += q; a += rem;
      if (a >= per) { a -= per; ++r; }
    }
#pragma unroll
    for (int k = 0; k < GN_SMALL_U; ++k) {
      if (u + k * GN_SMALL_NT < units) {
        buf[u + k * GN_SMALL_NT] = v[k];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float f = (float)v[k][e]; s += f; ss += f * f; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  if (lane == 0) { red[w] = (double)s; red[GN_SMALL_NT / 64 + w] = (double)ss; }
  __syncthreads();
  if (t == 0) {
    const double inv = 1.0 / ((double)rows * (double)gs);
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int i = 0; i < GN_SMALL_NT / 64; ++i) { sa += red[i]; sb += red[GN_SMALL_NT / 64 + i]; }
    const double mean = sa * inv;
    double var = sb * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[0] = (float)mean;
    mr[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (stats) { stats[(sample * GN_GROUPS + g) * 2] = mr[0]; stats[(sample * GN_GROUPS + g) * 2 + 1] = mr[1]; }
  }
  __syncthreads();
  const float mean = mr[0], rstd = mr[1];
  r = t / per; a = t - r * per;
  for (int u = t; u < units; u += GN_SMALL_NT) {
    const int c = g * gs + a * VEC;
    const hv_t v = buf[u];
    hv_t o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float A = rstd * gamma[c + e], B = beta[c + e] - mean * A;       // (the same A, B as gn_apply_kernel)
      float f = (float)v[e] * A + B;
      if (silu) f = silu_f(f);
      o[e] = (half_t)f;
    }
    *(hv_t*)(out + (base + r) * ldo + c) = o;
    r += q; a += rem;
    if (a >= per) { a -= per; ++r; }
  }
}

static thread_local int gn_small_on = 1;                 // A/B knob: 0 = always the three launches
extern "C" void lkgd_debug_set_gn_small(int on) { gn_small_on = on != 0; }
#define GN_SMALL_LDS (48 * 1024)
static thread_local long long gn_small_max_bytes = 10LL * 1024 * 1024 + 512 * 1024;
extern "C" void lkgd_debug_set_gn_small_limits(int64_t total_bytes) { gn_small_max_bytes = total_bytes; }
// does the one-launch form apply?  A workgroup reads its group as C/32-channel pieces of every row (20-160 bytes of each
// 128-byte line: ~1.3 TB/s chip-wide against the chunked passes' whole rows), so it pays only while the launch overheads of
// the three passes outweigh that: groups up to 48 KiB AND tensors up to 10.5 MB (tools/micro/gn_small.py, profiles/
// r05_gn_small.txt: 4 x 576 x 1280 14.1 vs 18.1 us, 28 x 144 x 1280 15.1 vs 18.0, 4 x 144 x 2560 13.1 vs 22.4; beyond -
// 14 x 576 x 1280 27.2 vs 22.5, 4 x 576 x 1920 22.8 vs 18.6 - the three launches win); at least 64 (sample, group) pairs
static int gn_small_vec(int c0, int c1, int ld0, int ld1, int ldo, long long nsamples, long long rows, const void* out) {
  const int C = c0 + c1, gs = C / GN_GROUPS;
  if (!gn_small_on || rows * gs * 2 > GN_SMALL_LDS || nsamples * rows * C * 2 > gn_small_max_bytes || nsamples * GN_GROUPS < 64)
    return 0;
  const int vec = gs % 8 == 0 ? 8 : (gs % 4 == 0 ? 4 : (gs % 2 == 0 ? 2 : 0));
  if (!vec || gs / vec > GN_SMALL_NT || c0 % vec || ld0 % vec || (c1 > 0 && ld1 % vec) || ldo % vec) return 0;
  if (((uintptr_t)out) % (2 * vec)) return 0;
  return vec;
}

// The whole GroupNorm (+ SiLU) of a tensor in one C call: one launch where a (sample, group) fits a workgroup's LDS
// (gn_small_kernel), else statistics pass, finalize, apply pass (three launches; what a caller saves there is two trips
// through its own call layer)
extern "C" int lkgd_groupnorm_silu(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                                   int64_t nsamples, int64_t rows_per_sample, float eps, float* partial, float* stats,
                                   const float* gamma, const float* beta, int32_t silu, void* out, int32_t ldo,
                                   lkgd_stream_t stream) {
  const int vec = gn_small_vec(c0, c1, ld0, ld1, ldo, nsamples, rows_per_sample, out);
  if (vec) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample);
    if (rc) return rc;
    if (!gamma || !beta || !out) return LKGD_E_NULL;
    LKGD_DEVICE_ONCE_BEGIN
      if (hipFuncSetAttribute((const void*)gn_small_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, GN_SMALL_LDS) != hipSuccess ||
          hipFuncSetAttribute((const void*)gn_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, GN_SMALL_LDS) != hipSuccess ||
          hipFuncSetAttribute((const void*)gn_small_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GN_SMALL_LDS) != hipSuccess)
        return LKGD_E_LAUNCH;
    LKGD_DEVICE_ONCE_END
    const size_t lds = (size_t)rows_per_sample * ((c0 + c1) / GN_GROUPS) * 2;
    const dim3 grid(GN_GROUPS, (unsigned)nsamples);
#define GN_SMALL_GO(V)                                                                                                   \
  hipLaunchKernelGGL(gn_small_kernel<V>, grid, dim3(GN_SMALL_NT), lds, (hipStream_t)stream, (const half_t*)x0, c0, ld0,          \
                     (const half_t*)x1, c1, ld1, (int)rows_per_sample, eps, gamma, beta, silu, (half_t*)out, ldo, stats)
    if (vec == 8) GN_SMALL_GO(8); else if (vec == 4) GN_SMALL_GO(4); else GN_SMALL_GO(2);
#undef GN_SMALL_GO
    return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
  }
  int rc = lkgd_groupnorm_stats(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample, eps, partial, stats, stream);
  if (rc) return rc;
  return lkgd_groupnorm_apply(x0, c0, ld0, x1, c1, ld1, nsamples, rows_per_sample, stats, gamma, beta, silu, out, ldo, stream);
}

// ---------------------------------------------------------------------------------------------------- LayerNorm
// A row is handled by a group of L lanes (L = 4..64, power of two) with up to 3 x 16-byte vectors per lane, so that small
// channel counts keep the lanes busy: C = 320 -> 16 lanes x 3 vectors, 4 rows per wave (83 % lane efficiency instead of
// 62 % with one row per wave); reductions are log2(L) shuffle steps inside the group.  gamma/beta stay in registers.
#define LN_MAXV 3          // 16-byte vectors per lane: rows up to 64 * 8 * 3 = 1536 channels
#define LN_MAXV_WIDE 4     // the DiT's 1920-channel rows (L = 64 lanes x 4 vectors = 2048)
template <int L, bool AFF, int NV = LN_MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* x, int ldx, long long T, int C,
                                                        const float* gamma, const float* beta, float eps,
                                                        const half_t* rowbias, int ldrb, int d1, int m1, int d2,
                                                        int md, int c0, half_t* out, int ldo) {
  constexpr int RPW = 64 / L;                       // rows per wave
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / L, li = lane % L;          // row slot inside the wave, lane inside the row group
  const int C8 = C >> 3;
  float g[AFF ? NV : 1][8], b[AFF ? NV : 1][8];
  if (AFF) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int cv = li + L * v;
      if (cv >= C8) cv = C8 - 1;        // clamped lanes never store
      const float4_t g0 = *(const float4_t*)(gamma + cv * 8), g1 = *(const float4_t*)(gamma + cv * 8 + 4);
      const float4_t b0 = *(const float4_t*)(beta + cv * 8), b1 = *(const float4_t*)(beta + cv * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        g[AFF ? v : 0][e] = e < 4 ? g0[e & 3] : g1[e & 3];
        b[AFF ? v : 0][e] = e < 4 ? b0[e & 3] : b1[e & 3];
      }
    }
  }
  const float invC = 1.0f / (float)C;
  const long long rows_per_block = 4 * RPW;
  for (long long row0 = (long long)blockIdx.x * rows_per_block + wave * RPW; row0 < T;
       row0 += (long long)gridDim.x * rows_per_block) {
    const long long row = row0 + sub;
    const bool live = row < T;
    float xv[NV][8];
    float s = 0.f;
    long long idx = 0;
    if (rowbias && live) {
      if (T < (1LL << 31)) {   // the usual case: 32-bit unsigned arithmetic (64-bit divisions cost more than the norm itself)
        const unsigned ru = (unsigned)row;
        idx = ((ru / (unsigned)d1) * (unsigned)m1 + (ru % (unsigned)d2) + (unsigned)c0) % (unsigned)md;
      } else {
        idx = ((row / d1) * m1 + (row % d2) + c0) % md;
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int cv = li + L * v;
      if (live && cv < C8) {
        half8_t h = *(const half8_t*)(x + row * ldx + cv * 8);
        if (rowbias) {
          half8_t rb = *(const half8_t*)(rowbias + idx * ldrb + cv * 8);
          h = h + rb;      // the reference adds in fp16 (hidden_states_mix + emb) before the norm
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[v][e] = (float)h[e]; s += xv[v][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[v][e] = 0.f;
      }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int cv = li + L * v;
      if (cv < C8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = xv[v][e] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * invC + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int cv = li + L * v;
      if (live && cv < C8) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = (xv[v][e] - mean) * rstd;
          if (AFF) y = y * g[AFF ? v : 0][e] + b[AFF ? v : 0][e];
          o[e] = (half_t)y;
        }
        *(half8_t*)(out + row * ldo + cv * 8) = o;
      }
    }
  }
}

extern "C" int lkgd_layernorm(const void* x, int32_t ldx, int64_t T, int32_t C, const float* gamma,
                              const float* beta, float eps, const void* rowbias, int32_t ldrb, int32_t rb_d1,
                              int32_t rb_m1, int32_t rb_d2, int32_t rb_md, void* out, int32_t ldo,
                              lkgd_stream_t stream) {
  if (!x || !out) return LKGD_E_NULL;
  if ((gamma == nullptr) != (beta == nullptr)) return LKGD_E_NULL;     // both or neither (neither = no affine)
  if (T <= 0 || C <= 0 || C % 8 || C > 64 * 8 * LN_MAXV_WIDE) return LKGD_E_SHAPE;
  if (ldx % 8 || ldo % 8 || !aligned16(x) || !aligned16(out)) return LKGD_E_ALIGN;
  if (rowbias && (ldrb % 8 || !aligned16(rowbias) || rb_d1 <= 0 || rb_d2 <= 0 || rb_md <= 0)) return LKGD_E_SHAPE;
  const int C8 = C / 8;
  int L = 4;
  while (L < 64 && L * LN_MAXV < C8) L *= 2;
  const long long rows_per_block = 4 * (64 / L);
  long long blocks = (T + rows_per_block - 1) / rows_per_block;
  if (blocks > 256 * 16) blocks = 256 * 16;
#define LN_LAUNCH(LL)                                                                                            \
  if (gamma)                                                                                                     \
    hipLaunchKernelGGL((layernorm_kernel<LL, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,  \
                       (const half_t*)x, ldx, (long long)T, C, gamma, beta, eps, (const half_t*)rowbias, ldrb,   \
                       rb_d1, rb_m1, rb_d2, rb_md, 0, (half_t*)out, ldo);                                        \
  else                                                                                                           \
    hipLaunchKernelGGL((layernorm_kernel<LL, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, \
                       (const half_t*)x, ldx, (long long)T, C, gamma, beta, eps, (const half_t*)rowbias, ldrb,   \
                       rb_d1, rb_m1, rb_d2, rb_md, 0, (half_t*)out, ldo)
  if (C8 > 64 * LN_MAXV) {          // 1536 < C <= 2048: one row per wave, four vectors per lane
    if (gamma)
      hipLaunchKernelGGL((layernorm_kernel<64, true, LN_MAXV_WIDE>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                         (const half_t*)x, ldx, (long long)T, C, gamma, beta, eps, (const half_t*)rowbias, ldrb, rb_d1, rb_m1,
                         rb_d2, rb_md, 0, (half_t*)out, ldo);
    else
      hipLaunchKernelGGL((layernorm_kernel<64, false, LN_MAXV_WIDE>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                         (const half_t*)x, ldx, (long long)T, C, gamma, beta, eps, (const half_t*)rowbias, ldrb, rb_d1, rb_m1,
                         rb_d2, rb_md, 0, (half_t*)out, ldo);
    return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
  }
  switch (L) {
    case 4: LN_LAUNCH(4); break;
    case 8: LN_LAUNCH(8); break;
    case 16: LN_LAUNCH(16); break;
    case 32: LN_LAUNCH(32); break;
    default: LN_LAUNCH(64); break;
  }
#undef LN_LAUNCH
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
