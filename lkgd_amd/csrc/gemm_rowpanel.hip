// "Row-panel" MFMA GEMM for short K (K <= 320): out[M, N] = x[M, K] . W[N, K]^T with the fused epilogue of
// include/lkgd_hip.h section 1 (plain A operand only).
//
// Why a separate kernel: at K = 320 (every projection of the 72x128 level: QKV, attention out, proj_in/out, GEGLU) a
// tiled GEMM re-reads the token tile for every column tile; the CU's L2 -> LDS path (~29 B/cycle) then bounds the kernel
// at ~550 TFLOP/s (profiles/r01_gemm_shapes.txt, tools/gemm_phase_stamps.py).  Here the token panel never re-enters the
// CU:
//   * a workgroup (8 waves) owns 256 token rows; each wave keeps ITS 32 rows x K as MFMA B-operand fragments in
//     registers for the whole sweep over N (K = 320: 20 fragments = 80 VGPRs) - loaded once from HBM;
//   * the weight matrix streams through a 3-buffer LDS ring in WHOLE column tiles (64 packed rows x K, 40 KiB at
//     K = 320) by LDS-DMA, shared by the 8 waves: 16 B/cycle/CU of load traffic instead of 47;
//   * one barrier per column tile; the ring wait is taken right before the epilogue, where the only outstanding vector
//     memory operations are the two tiles in flight, so epilogue loads/stores never disturb the counted wait;
//   * accumulator = [channel][token] (weights as MFMA A operand): lane = token, 4 consecutive channels per register
//     group; rows leave through a per-wave LDS transpose as 16-byte coalesced stores (residual rows arrive the same way);
//   * persistent workgroups; adjacent CUs take adjacent row panels and all stream the same (L2-resident) weights.
#include "gemm_common.h"

#define RP_ROWS 256          // token rows per workgroup
#define RP_BN 64             // packed weight rows (output channels) per tile
#define RP_NT 512
#define RP_NBUF 3

__device__ __forceinline__ float gelu_fast_rp(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}

#ifdef LKGD_GEMM_STAMPS
extern __device__ unsigned long long lkgd_gemm_stamps[256 * 8];
#define RSTAMP(var) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; }
#else
#define RSTAMP(var)
#endif

// NK = K / 64 (1..5).  LDS: RP_NBUF weight tiles of 64 x K fp16 + 8 per-wave transpose scratches of 32 x 144 B.
// LNF: LayerNorm of the token rows folded in (lkgd_gemm_desc.ln_colsum): the rows sit in registers for the whole sweep over N, so
// their mean / rstd cost one pass of v_dot2 over the fragments per panel, and the epilogue applies
// rstd * (acc - mean * colsum[n]) before the bias - the normalised rows are never materialised (fp32 arithmetic on exact fp16
// products: more accurate than rounding the normalised row to fp16 first).
template <int NK, bool LNF>
__global__ __launch_bounds__(RP_NT, 2) void lkgd_gemm_rowpanel_kernel(const lkgd_gemm_desc p, int panels, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = NK * 64;
  constexpr int CPR = NK * 8;                      // 16-byte chunks per weight row
  constexpr int TILE_BYTES = RP_BN * K * 2;
  constexpr int NCH = RP_BN * CPR;                 // chunks per weight tile
  constexpr int NLD = (NCH + RP_NT - 1) / RP_NT;   // LDS-DMA ops per thread per tile
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  char* scr = smem + RP_NBUF * TILE_BYTES + w * 4608;

  // ---- persistent schedule over row panels: XCD-cooperative round-robin (adjacent CUs, adjacent panels)
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  const int nc = (G - xcd + 7) >> 3;
  const int q8 = panels >> 3, r8 = panels & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);
  const int xe = xb + q8 + (xcd < r8 ? 1 : 0);
  const int my_panels = (xe - xb - c + nc - 1) / nc;
  if (my_panels <= 0) return;
  const int total = my_panels * tiles_n;           // weight tiles in this workgroup's stream

  // ---- weight staging map: thread fills LDS slots t + 512*i of a tile; slot -> (row, physical chunk); the logical
  //      chunk it fetches is XOR-swizzled inside each 128-byte segment (conflict-free ds_read_b128 of fragments)
  int w_src[NLD];                                  // element offset of the source chunk inside a weight tile
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int q = t + RP_NT * i;
    const int row = q / CPR, pc = q - row * CPR;
    const int lc = (pc & ~7) | ((pc & 7) ^ ((row >> 1) & 7));
    w_src[i] = q < NCH ? row * K + lc * 8 : -1;
  }
  int st = 0;                                      // next tile of the stream to stage
  auto stage = [&]() {
    const int tn = st % tiles_n;
    char* dst = smem + (st % RP_NBUF) * TILE_BYTES;
    const half_t* wt = (const half_t*)p.w + (long long)tn * RP_BN * K;
    const int nrow0 = tn * RP_BN;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (w_src[i] >= 0) {
        const int row = (t + RP_NT * i) / CPR;
        const half_t* src = nrow0 + row < p.N ? wt + w_src[i] : (const half_t*)p.zeros;
        glds16(src, dst + (w * 64 + RP_NT * i) * 16);
      }
    }
    ++st;
  };

  // fragment read offsets inside a weight tile: row (i*32 + l31), chunk (ks*2 + h), swizzled
  const int skey = (l31 >> 1) & 7;
  const int wrow_off = l31 * (K * 2);

  stage();
  if (total > 1) stage();

  half8_t xq[NK * 4];
  float16_t acc[2];
  float ln_mean = 0.f, ln_rstd = 1.f;              // LNF: statistics of this lane's token row
  float4_t ln_cs[8];                               // LNF: column sums of the current weight tile (this lane's 8 channel quads)
  int tile_in_panel = 0, panel_idx = 0;
  long long m_row = 0;
  // Optional stagger (waves 4-7 run the epilogue of tile s-1 right after the barrier of step s while waves 0-3 stage +
  // compute tile s).  Measured SLOWER here (GEGLU 2560x320 @ 258k rows: 10.1 ms vs 9.15 ms unstaggered), unlike in
  // gemm_stream.hip - kept compiled out.
  constexpr bool kStagger = false;
  const bool late = kStagger && __builtin_amdgcn_readfirstlane(t) >= 256;
  int e_tn = 0;
  long long e_mb = 0, e_mrow = 0;
  bool e_pending = false;
  // ---- epilogue of one 32-token x 64-channel wave tile, straight from the accumulators
  auto epilogue = [&](const int tn, const long long mb, const long long mrow) {
    // ------------------------------------------------------------------------------------------ epilogue (registers)
    const int n0 = tn * RP_BN;
    const half_t* rbp = (const half_t*)p.rowbias;
    const half_t* r1p = (const half_t*)p.res1;
    const half_t* r2p = (const half_t*)p.res2;
    half_t* outp = (half_t*)p.out;
    if (!p.geglu) {
      const int crow = lane >> 3, cchunk = lane & 7;              // coalesced map: 8 lanes x 16 B per 128-byte row
      const int ncol = n0 + cchunk * 8;
      uint4 rres[4];
      if (r1p) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const long long mr = mb + crow + 8 * it;
          rres[it] = (mr < p.M && ncol < p.N) ? *(const uint4*)(r1p + mr * p.ldr1 + ncol) : uint4{0u, 0u, 0u, 0u};
        }
      }
      long long idx = 0;
      if (rbp && mrow < p.M) idx = (((unsigned)mrow / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)mrow % (unsigned)p.rb_d2) + (unsigned)p.rb_c0) % (unsigned)p.rb_md;   // M is an int32
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + i * 32 + 8 * g + 4 * h;
          float4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
          if (n < p.N) {
            if (LNF) v = (v - ln_mean * ln_cs[i * 4 + g]) * ln_rstd;
            if (p.bias) v += *(const float4_t*)(p.bias + n);
            if (rbp && mrow < p.M) {
              half4_t rb = *(const half4_t*)(rbp + idx * p.ldrb + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
            }
          }
          v *= p.s_acc;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = v[e];
        }
      if (r1p) {
#pragma unroll
        for (int it = 0; it < 4; ++it) *(uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16) = rres[it];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4_t r = *(const half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][4 * g + e] += p.r1 * (float)r[e];
          }
      }
      if (r2p) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const long long mr = mb + crow + 8 * it;
          uint4 x = {0u, 0u, 0u, 0u};
          if (mr < p.M && ncol < p.N) x = *(const uint4*)(r2p + mr * p.ldr2 + ncol);
          *(uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16) = x;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4_t r = *(const half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][4 * g + e] += p.r2 * (float)r[e];
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)acc[i][4 * g + e];
          *(half4_t*)(scr + l31 * 144 + (i * 32 + 8 * g + 4 * h) * 2) = o;
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long long mr = mb + crow + 8 * it;
        const uint4 x = *(const uint4*)(scr + (crow + 8 * it) * 144 + cchunk * 16);
        if (mr < p.M && ncol < p.N) *(uint4*)(outp + mr * p.ldc + ncol) = x;
      }
    } else {
      // packed rows of this tile: [32 hidden | 32 gate] of output columns tn*32 + [0,32)
      const int oc0 = tn * 32;
      const int grow = lane >> 2, gchunk = lane & 3;              // 4 lanes x 16 B per 64-byte row
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = 8 * g + 4 * h;
        float4_t bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          bh = *(const float4_t*)(p.bias + n0 + cc);
          bg = *(const float4_t*)(p.bias + n0 + 32 + cc);
        }
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = (half_t)((acc[0][4 * g + e] + bh[e]) * gelu_fast_rp(acc[1][4 * g + e] + bg[e]));
        *(half4_t*)(scr + l31 * 144 + cc * 2) = o;
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const long long mr = mb + grow + 16 * it;
        const uint4 x = *(const uint4*)(scr + (grow + 16 * it) * 144 + gchunk * 16);
        if (mr < p.M) *(uint4*)(outp + mr * p.ldc + oc0 + gchunk * 8) = x;
      }
    }
  };


#ifdef LKGD_GEMM_STAMPS
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, b_wait = 0, b_stage = 0, b_comp = 0, b_epi = 0, q_begin = 0;
  RSTAMP(q_begin)
#endif
  for (int s = 0; s < total; ++s) {
    RSTAMP(q0)
    if (tile_in_panel == 0) {
      // ---- new row panel: this wave's 32 token rows x K into registers (B-operand fragments)
      const int panel = xb + c + panel_idx * nc;
      m_row = (long long)panel * RP_ROWS + w * 32 + l31;
      const half_t* xp = (const half_t*)p.a0 + (m_row < p.M ? m_row : 0) * p.lda0 + h * 8;
#pragma unroll
      for (int ks = 0; ks < NK * 4; ++ks) {
        half8_t v = *(const half8_t*)(xp + ks * 16);
        if (m_row >= p.M) v = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        xq[ks] = v;
      }
      // these loads are older than nothing the ring wait below depends on: tile s was issued before them
    }
    // ---- tile s must have landed.  At this point of the wave's queue only ring traffic for tiles s (and s+1) can be
    //      older than what we leave outstanding; the x-panel loads above are YOUNGER than tile s (+ tile s+1).
    if (s == 0 || tile_in_panel == 0) {
      // queue may hold [tile s][tile s+1][x panel loads]: wait for everything (once per panel)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (LNF && tile_in_panel == 0) {
        // this lane holds half of its row (k = 16 ks + 8 h ..): sum and sum of squares by v_dot2, the other half one lane away
        float s1 = 0.f, s2 = 0.f;
        const half2_t one = {(half_t)1.f, (half_t)1.f};
#pragma unroll
        for (int ks = 0; ks < NK * 4; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2_t u = {xq[ks][2 * e], xq[ks][2 * e + 1]};
            s1 = __builtin_amdgcn_fdot2(u, one, s1, false);
            s2 = __builtin_amdgcn_fdot2(u, u, s2, false);
          }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        ln_mean = s1 * (1.0f / K);
        float var = s2 * (1.0f / K) - ln_mean * ln_mean;
        var = var < 0.f ? 0.f : var;
        ln_rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
      }
    }
    __builtin_amdgcn_s_barrier();
    RSTAMP(q1)
    if (LNF) {
      // this tile's column sums, issued BEFORE the next ring stage so that the counted ring wait below retires them with tile
      // s+1 (queue: [tile s+1][these][tile s+2]) and the epilogue finds them in registers instead of waiting for L2
      const int n0 = tile_in_panel * RP_BN;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + i * 32 + 8 * g + 4 * h;
          ln_cs[i * 4 + g] = n < p.N ? *(const float4_t*)(p.ln_colsum + n) : (float4_t){0.f, 0.f, 0.f, 0.f};
        }
      asm volatile("" ::: "memory");
    }
    if (s + 2 < total) stage();
    if (late) {
      // late group: its ring wait for tile s+1 sits here, where its queue is [tile s+1][epilogue s-2][tile s+2]
      if (s + 1 < total) {
        if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (e_pending) { epilogue(e_tn, e_mb, e_mrow); e_pending = false; }
    }
    RSTAMP(q2)
    const char* wb = smem + (s % RP_NBUF) * TILE_BYTES + wrow_off;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // weight fragments (LDS reads) run 6 ahead of the MFMA that consumes them: the scheduler groups below pin the
    // issue order [6 reads] then [MFMA, read] pairs, so each MFMA only waits for a read issued six slots earlier
    {
      constexpr int NF = NK * 8;                       // fragment reads of this tile: (ks, i) -> f = ks*2 + i
      half8_t wf[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int ks = f >> 1, i = f & 1;
        const int ch = ks * 2 + h;
        wf[f] = *(const half8_t*)(wb + i * 32 * (K * 2) + (((ch & ~7) | ((ch & 7) ^ skey)) << 4));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f)
        acc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[f], xq[f >> 1], acc[f & 1], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NF < 6 ? NF : 6, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (f + 6 < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    // ---- take the NEXT step's ring wait now: outstanding = [tile s+1][tile s+2] (+ nothing else): "at most NLD"
    //      retires tile s+1 before any epilogue traffic enters the queue
    if (s + 1 < total) {
      if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    RSTAMP(q3)
    {
      const long long mb_now = (long long)(xb + c + panel_idx * nc) * RP_ROWS + w * 32;   // first token row of this wave
      if (!late) epilogue(tile_in_panel, mb_now, m_row);
      else { e_tn = tile_in_panel; e_mb = mb_now; e_mrow = m_row; e_pending = true; }
    }
    if (++tile_in_panel == tiles_n) { tile_in_panel = 0; ++panel_idx; }
    RSTAMP(q4)
#ifdef LKGD_GEMM_STAMPS
    b_wait += q1 - q0; b_stage += q2 - q1; b_comp += q3 - q2; b_epi += q4 - q3;
#endif
  }
  if (late && e_pending) epilogue(e_tn, e_mb, e_mrow);
#ifdef LKGD_GEMM_STAMPS
  if (t == 0 && blockIdx.x < 256) {
    unsigned long long q_end; RSTAMP(q_end)
    unsigned long long* o = lkgd_gemm_stamps + blockIdx.x * 8;
    o[0] = b_wait; o[1] = b_stage; o[2] = b_comp; o[3] = b_epi; o[4] = q_end - q_begin; o[5] = total;
  }
#endif
}

extern "C" int lkgd_gemm_rowpanel_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  const int nk = d->K / 64;
  if (nk < 1 || nk > 5 || d->mode != LKGD_A_PLAIN || d->csplit < d->K) return LKGD_E_SHAPE;
  if (d->ln_colsum && (d->geglu || d->K != nk * 64)) return LKGD_E_SHAPE;
  const int lds = RP_NBUF * RP_BN * d->K * 2 + 8 * 4608;
  LKGD_DEVICE_ONCE_BEGIN
    const int mx = RP_NBUF * RP_BN * 320 * 2 + 8 * 4608;
#define RP_ATTR(NKV, L) (hipFuncSetAttribute((const void*)lkgd_gemm_rowpanel_kernel<NKV, L>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
    if (RP_ATTR(1, false) || RP_ATTR(2, false) || RP_ATTR(3, false) || RP_ATTR(4, false) || RP_ATTR(5, false) ||
        RP_ATTR(3, true) || RP_ATTR(4, true) || RP_ATTR(5, true))
      return LKGD_E_LAUNCH;
#undef RP_ATTR
  LKGD_DEVICE_ONCE_END
  const int panels = (d->M + RP_ROWS - 1) / RP_ROWS;
  const int tiles_n = (d->N + RP_BN - 1) / RP_BN;
  const int grid = panels < cus ? panels : cus;
#define RP_LAUNCH(NKV)                                                                                       \
  hipLaunchKernelGGL((lkgd_gemm_rowpanel_kernel<NKV, false>), dim3(grid), dim3(RP_NT), lds, stream, *d, panels, tiles_n)
#define RP_LAUNCH_LN(NKV)                                                                                    \
  hipLaunchKernelGGL((lkgd_gemm_rowpanel_kernel<NKV, true>), dim3(grid), dim3(RP_NT), lds, stream, *d, panels, tiles_n)
  if (d->ln_colsum) {
    switch (nk) {       // the LayerNorm widths of the model's K <= 320 projections: 192 (tiny configs), 256, 320
      case 3: RP_LAUNCH_LN(3); break;
      case 4: RP_LAUNCH_LN(4); break;
      case 5: RP_LAUNCH_LN(5); break;
      default: return LKGD_E_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
  }
  switch (nk) {
    case 1: RP_LAUNCH(1); break;
    case 2: RP_LAUNCH(2); break;
    case 3: RP_LAUNCH(3); break;
    case 4: RP_LAUNCH(4); break;
    default: RP_LAUNCH(5); break;
  }
#undef RP_LAUNCH
#undef RP_LAUNCH_LN
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
