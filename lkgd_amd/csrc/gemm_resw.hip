// "Resident-weight" MFMA GEMM for short K (K <= 320): out[M, N] = x[M, K] . W[N, K]^T with the fused epilogue of
// include/lkgd_hip.h section 1 (plain A operand, N a multiple of 160).
//
// Why another tile program: the K = 320 projections of the 72x128 level (QKV, attention out, proj_in / proj_out, GEGLU;
// reference call sites patch/patch.py:440-445, 543-580) spend as long in their epilogue (erf, residual rows, 0.5-0.7 GB of
// stores) as in their K-loop, and in the barrier-coupled tile kernels all eight waves of a CU are in the epilogue together:
// the matrix pipe idles, the store path is hit by every CU at once (profiles/r02_gemm_store_knobs.txt).  A second
// accumulator set does not fit beside a 256x320 tile (320 KB of the CU's 512 KB register file).  What does fit at K <= 320
// is the OTHER operand: a 160-channel slab of the weights is 100 KB and stays in LDS for the workgroup's whole life, so
//   * nothing is staged per tile: no LDS-DMA stream, no barrier after the prologue - the eight waves run independently,
//     and while one wave of a SIMD converts / stores its rows the other one's MFMAs have the matrix pipe;
//   * a wave owns 32 token rows at a time, held as MFMA B-operand fragments in registers (K = 320: 80 VGPRs), loaded
//     straight from HBM with 16-byte loads; the next block's fragments are loaded into the registers of K-steps already
//     consumed, so the load latency hides behind the rest of the K-loop and the epilogue;
//   * weight fragments (v_mfma_f32_16x16x32_f16 A operands, 1 KiB each) sit fragment-major in LDS: every read is one
//     conflict-free ds_read_b128 at base + lane * 16, feeding two MFMAs (both token fragments);
//   * accumulator = [channel][token]: a lane holds 4 consecutive channels of one token row (the layout of gemm_wide.hip);
//     rows leave through a wave-private LDS patch as 16-byte pieces of whole 320-byte (GEGLU: 160-byte) row segments;
//   * workgroup c of an XCD (blocks b, b + 8, ... share one) keeps slab c % nslab and sweeps part c / nslab of that XCD's
//     rows: the nslab workgroups that read the same token rows run on one XCD and meet in its L2.
#include "gemm_common.h"

// timing-experiment knobs (tools/micro/resw_knobs.sh; results are wrong with any of them): never defined in the product build
#ifdef RESW_X_NOREAD
#define RESW_RD(X) ""
#else
#define RESW_RD(X) X
#endif
#ifdef RESW_X_NOMFMA
#define RESW_MM(X) ""
#else
#define RESW_MM(X) X
#endif
#ifdef RESW_X_NOLOADX
#define RESW_LX(X) ""
#else
#define RESW_LX(X) X
#endif
#include "gemm_resw_kloop.inc"

#ifdef RESW_X_STAMPS        /* diagnostic build: per-wave cycle sums of the K-loop / epilogue arithmetic / row stores */
__device__ unsigned long long lkgd_resw_stamps[256 * 8 * 4];
extern "C" int lkgd_debug_resw_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lkgd_resw_stamps), (size_t)n * 8);
}
#define RSTAMP(var) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; }
#else
#define RSTAMP(var)
#endif

#define RW_SLAB 160
#define RW_NT 512
#define RW_WAVES 8
#define RW_PITCH 336           // bytes per staged output row (320 + 16: rows start on different banks)
#define RW_STAGE (16 * RW_PITCH)

// accumulator fragment (channel fragment i, token fragment j) out of the AGPRs; BASE = (2i + j) * 4
template <int BASE>
__device__ __forceinline__ float4_t resw_read_acc() {
  float a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\t"
               "v_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
               : "i"(BASE));
  return (float4_t){a, b, c, d};
}

// 16-byte store of one row piece, issued by every lane's wave whatever the mask (the K-loop's counted vmcnt relies on
// the number of store instructions between two K-loops); lanes whose row is past M are masked off
__device__ __forceinline__ void resw_store16(half_t* ptr, half8_t v, int mr, int M) {
  unsigned long long sv;
#ifdef RESW_X_NOSTORE       /* every store instruction issued with all lanes masked off */
  M = 0;
#endif
  // (a VALU write of the data registers needs two wait states behind a 16-byte store on gfx940+; the compiler's hazard
  // recognizer does not look into inline asm, so the statement ends with them)
  asm volatile("v_cmp_gt_i32 vcc, %3, %4\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dwordx4 %1, %2, off\n\ts_mov_b64 exec, %0\n\ts_nop 1"
               : "=&s"(sv)
               : "v"(ptr), "v"(v), "s"(M), "v"(mr)
               : "vcc", "memory");
}

template <int V> struct ReswIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void resw_static_for(F&& f, ReswIC<Is>...) { (f(ReswIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void resw_for10(F&& f) {
  resw_static_for(f, ReswIC<0>{}, ReswIC<1>{}, ReswIC<2>{}, ReswIC<3>{}, ReswIC<4>{}, ReswIC<5>{}, ReswIC<6>{}, ReswIC<7>{}, ReswIC<8>{}, ReswIC<9>{});
}
template <class F> __device__ __forceinline__ void resw_for5(F&& f) {
  resw_static_for(f, ReswIC<0>{}, ReswIC<1>{}, ReswIC<2>{}, ReswIC<3>{}, ReswIC<4>{});
}

// K-loop of one block (generated asm, tools/gen_resw_asm.py).  YOUNGER = vector-memory operations that are certainly
// issued between the previous K-loop's last token-fragment load and this K-loop: the epilogue's row-piece stores
// (resw_store16: 10, GEGLU 6).  vmcnt counts loads and stores together, in issue order, so "all but the 2*NKS - 2 +
// YOUNGER youngest" retires K-step 0's fragments without waiting for the stores behind them; a LOWER bound is safe
// (more operations in flight than assumed only makes the wait stricter).
template <int NK, int YOUNGER>
__device__ __forceinline__ void resw_kloop(int wl0, int wl1, const half_t* pn0, const half_t* pn1) {
  half8_t f0, f1, f2, f3, f4, f5, f6, f7, f8, f9;
  constexpr int W0 = 4 * NK - 2 + YOUNGER, W1 = 4 * NK - 4 + YOUNGER;
#define RESW_STMT(BODY)                                                                                   \
  asm volatile(BODY                                                                                       \
               : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7),  \
                 "=&v"(f8), "=&v"(f9)                                                                     \
               : "v"(wl0), "v"(wl1), "v"(pn0), "v"(pn1), "n"(W0 < 63 ? W0 : 63), "n"(W1 < 63 ? W1 : 63)   \
               : "memory", RESW_AGPR_CLOBBERS)
  if (NK == 1) RESW_STMT(RESW_KLOOP_ASM_1);
  else if (NK == 2) RESW_STMT(RESW_KLOOP_ASM_2);
  else if (NK == 3) RESW_STMT(RESW_KLOOP_ASM_3);
  else if (NK == 4) RESW_STMT(RESW_KLOOP_ASM_4);
  else RESW_STMT(RESW_KLOOP_ASM_5);
#undef RESW_STMT
}

template <int NK>
__device__ __forceinline__ void resw_loadx(const half_t* p0, const half_t* p1) {
#define RESW_STMT(BODY) asm volatile(BODY : : "v"(p0), "v"(p1) : "memory", RESW_AGPR_CLOBBERS)
  if (NK == 1) RESW_STMT(RESW_LOADX_ASM_1);
  else if (NK == 2) RESW_STMT(RESW_LOADX_ASM_2);
  else if (NK == 3) RESW_STMT(RESW_LOADX_ASM_3);
  else if (NK == 4) RESW_STMT(RESW_LOADX_ASM_4);
  else RESW_STMT(RESW_LOADX_ASM_5);
#undef RESW_STMT
}

// residual / row-bias values of one epilogue piece (token fragment j, channel fragments 5h .. 5h+4), in the accumulator
// layout (8 bytes per lane and fragment).  Loaded and awaited by hand: the compiler does not know about the token
// fragments in flight in the AGPRs and would drain them (and the previous rows' stores) with every wait it places.
struct ReswPiece { half4_t rb[5], r1[5], r2[5]; };

template <int I5>
__device__ __forceinline__ half4_t resw_ld8(const half_t* row) {
  half4_t v;
  asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v) : "v"(row), "n"(I5 * 32) : "memory");
  return v;
}
// wait until at most N vector-memory operations are outstanding, and tie the piece's registers to the wait (the
// compiler must not read them before it); only the sources present are tied - absent ones cost no registers
__device__ __forceinline__ void resw_tie5(half4_t (&a)[5]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
}
template <int N, int SRC>
__device__ __forceinline__ void resw_wait(ReswPiece& c) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
  if (SRC & 1) resw_tie5(c.rb);
  if (SRC & 2) resw_tie5(c.r1);
  if (SRC & 4) resw_tie5(c.r2);
}

// SRC: bit 0 = row-indexed bias, bit 1 = res1, bit 2 = res2 present (plain epilogue only); CS: column statistics
// (lkgd_gemm_desc.colstats) are compiled in - only for the source sets that keep the register budget (resw_has_cs)
template <int NK, bool GEGLU, int SRC, bool CS>
__global__ __launch_bounds__(RW_NT, 2) __attribute__((amdgpu_num_vgpr(96))) void lkgd_gemm_resw_kernel(const lkgd_gemm_desc p, int nslab, int per) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = NK * 2;                  // K-steps of 32
  constexpr int K = NK * 64;
  constexpr int WBYTES = 10 * NKS * 1024;      // 10 channel fragments x NKS K-steps x 1 KiB
  constexpr bool HAS_RB = (SRC & 1) != 0, HAS_R1 = (SRC & 2) != 0, HAS_R2 = (SRC & 4) != 0;
  constexpr int NSRC = (HAS_RB ? 1 : 0) + (HAS_R1 ? 1 : 0) + (HAS_R2 ? 1 : 0);
  constexpr int PL = 5 * NSRC;                 // loads per piece
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  if (c >= per * nslab) return;
  const int slab = c % nslab, part = c / nslab;

  // ---- the slab's weights, fragment-major: fragment f = i * NKS + ks = rows slab*160 + 16i + [0,16), K range 32ks + [0,32);
  //      lane (l15, lq) supplies row l15's 16 bytes at 8lq (the MFMA A-operand layout), LDS-DMA writes base + lane*16
  {
    const half_t* wsl = (const half_t*)p.w + (long long)slab * RW_SLAB * K + l15 * K + lq * 8;
#pragma unroll 1
    for (int f = w; f < 10 * NKS; f += RW_WAVES) {
      const int i = f / NKS, ks = f - i * NKS;
      glds16(wsl + i * 16 * K + ks * 32, smem + f * 1024);
    }
  }
  float* bsm = (float*)(smem + WBYTES);
  if (t < RW_SLAB) bsm[t] = p.bias ? p.bias[slab * RW_SLAB + t] : 0.f;
  char* scr = smem + WBYTES + RW_SLAB * 4 + w * RW_STAGE;

  // ---- this wave's 32-row blocks: the XCD's share of the blocks, this workgroup's part of it, every 8th block
  const int nb = (p.M + 31) >> 5;
  const int q8 = nb >> 3, r8 = nb & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);
  const int xn = q8 + (xcd < r8 ? 1 : 0);
  const int qp = xn / per, rp = xn - qp * per;
  const int pb = xb + part * qp + (part < rp ? part : rp);
  const int pe = pb + qp + (part < rp ? 1 : 0);

  // token row of fragment j of block b for this lane (rows past M re-read the last row; their results are never stored)
  const half_t* ap = (const half_t*)p.a0 + lq * 8;
  const unsigned lda = (unsigned)p.lda0;
  auto xrow = [&](int b, int j) {
    int m = b * 32 + j * 16 + l15;
    m = m < p.M ? m : p.M - 1;
#ifdef RESW_X_SMALLA        /* the token rows come from the first 4096 rows only: they stay L2-resident */
    m &= 4095;
#endif
#ifdef RESW_X_COALX         /* four consecutive lanes read 64 contiguous bytes of one row (a wrong operand, coalesced) */
    return (const half_t*)p.a0 + (unsigned long long)(unsigned)(b * 32 + j * 16 + (lane >> 2)) * lda + (lane & 3) * 8;
#endif
    return ap + (unsigned long long)(unsigned)m * lda;
  };
  int blk = pb + w;
  if (blk < pe) resw_loadx<NK>(xrow(blk, 0), xrow(blk, 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                // the slab and the bias strip are in LDS: the last barrier of the kernel

#ifdef RESW_X_STAGGER       /* timing knob: wave w starts w * RESW_X_STAGGER * 1024 cycles late */
  for (int i = 0; i < w * RESW_X_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif
  half_t* outp = (half_t*)p.out;
  const bool has_bias = p.bias != nullptr;
  const int wl0 = lane * 16, wl1 = lane * 16 + 5 * NKS * 1024;
  ReswPiece pc0, pc1;

#ifdef RESW_X_STAMPS
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, b_k = 0, b_e = 0, b_s = 0, b_n = 0;
#endif
#pragma unroll 1
  for (; blk < pe; blk += RW_WAVES) {
    RSTAMP(q0)
    const int nxt = blk + RW_WAVES < pe ? blk + RW_WAVES : blk;   // (the last block re-reads itself: uniform operation counts)
    // vector-memory operations between two K-loops: 4 pieces x PL loads + the row-piece stores (10, GEGLU 6)
    resw_kloop<NK, (GEGLU ? 6 : 10) + 4 * PL>(wl0, wl1, xrow(nxt, 0), xrow(nxt, 1));

    RSTAMP(q1)
    // ------------------------------------------------------------------------------------------------ epilogue
    int lane_e = lane;             // opaque copy: everything per-lane below is recomputed here, once per block, instead of
    asm volatile("" : "+v"(lane_e));   // being hoisted out of the loop into registers (then spilled: the budget is 96)
    const int lane = lane_e, l15 = lane_e & 15, lq = lane_e >> 4;      // (shadow the kernel-scope ones)
    if (!GEGLU) {
      // this lane's rows of the three sources for token fragment j (first channel of the slab + 4lq)
      const half_t *rbr[2], *r1r[2], *r2r[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int m = blk * 32 + j * 16 + l15;
        m = m < p.M ? m : p.M - 1;
        const int n = slab * RW_SLAB + 4 * lq;
        rbr[j] = r1r[j] = r2r[j] = nullptr;
        if (HAS_RB) {
          const unsigned idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) +
                                (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          rbr[j] = (const half_t*)p.rowbias + (unsigned long long)idx * (unsigned)p.ldrb + n;
        }
        if (HAS_R1) r1r[j] = (const half_t*)p.res1 + (unsigned long long)(unsigned)m * (unsigned)p.ldr1 + n;
        if (HAS_R2) r2r[j] = (const half_t*)p.res2 + (unsigned long long)(unsigned)m * (unsigned)p.ldr2 + n;
      }
      auto issue = [&](ReswPiece& pc, int j, auto hc) {
        constexpr int h = decltype(hc)::value;
        resw_for5([&](auto ic) {
          constexpr int i5 = decltype(ic)::value;
          if (HAS_RB) pc.rb[i5] = resw_ld8<h * 5 + i5>(rbr[j]);
          if (HAS_R1) pc.r1[i5] = resw_ld8<h * 5 + i5>(r1r[j]);
          if (HAS_R2) pc.r2[i5] = resw_ld8<h * 5 + i5>(r2r[j]);
        });
      };
      auto compute = [&](const ReswPiece& pc, auto jc, auto hc) {
        constexpr int j = decltype(jc)::value, h = decltype(hc)::value;
        float4_t bv[5];              // the piece's bias vectors in one LDS round trip
        if (has_bias) {
#pragma unroll
          for (int i5 = 0; i5 < 5; ++i5) bv[i5] = *(const float4_t*)(bsm + (h * 5 + i5) * 16 + 4 * lq);
        }
        resw_for5([&](auto ic) {
          constexpr int i5 = decltype(ic)::value, i = h * 5 + i5;
          float4_t v = resw_read_acc<(2 * i + j) * 4>();
          if (has_bias) v += bv[i5];
          if (HAS_RB) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] += (float)pc.rb[i5][x];
          }
          v *= p.s_acc;
          if (HAS_R1) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] += p.r1 * (float)pc.r1[i5][x];
          }
          if (HAS_R2) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] += p.r2 * (float)pc.r2[i5][x];
          }
          half4_t o;
#pragma unroll
          for (int x = 0; x < 4; ++x) o[x] = (half_t)v[x];
          *(half4_t*)(scr + l15 * RW_PITCH + (i * 16 + 4 * lq) * 2) = o;
        });
      };
      // the fragment's 16 rows as 16-byte pieces of 320-byte segments: piece q -> row q / 20, piece q % 20
      auto rows = [&](int j) {
        const int mrow0 = blk * 32 + j * 16;
        half8_t v[5];
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const int q = lane + 64 * kk;
          const int row = (q * 205) >> 12;            // q / 20 for q < 320
          v[kk] = *(const half8_t*)(scr + row * RW_PITCH + (q - row * 20) * 16);
        }
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const int q = lane + 64 * kk;
          const int row = (q * 205) >> 12;
          const int cc = q - row * 20;
          const int mr = mrow0 + row;
          resw_store16(outp + (unsigned long long)(unsigned)mr * (unsigned)p.ldc + slab * RW_SLAB + cc * 8, v[kk], mr, p.M);
        }
      };
      // column statistics of the block's rounded outputs (lkgd_gemm_desc.colstats, 32-row blocks): lanes 0-39 add four
      // channels each over the 16 row segments of a fragment while it sits in the patch
      const bool cs_on = CS && p.colstats != nullptr;
      float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};      // this lane's two channel PAIRS
      auto colsum = [&](int j) {
        if (CS && cs_on && lane < 40) {
          const int left = p.M - (blk * 32 + j * 16);
          const int nrows = left >= 16 ? 16 : (left > 0 ? left : 0);
          #pragma unroll
          for (int r0 = 0; r0 < 16; r0 += 8) {      // eight rows in flight at a time (16 registers)
            half4_t u[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) u[r] = *(const half4_t*)(scr + (r0 + r) * RW_PITCH + lane * 8);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const half_t one = r0 + r < nrows ? (half_t)1.0f : (half_t)0.0f;
              const half2_t w2 = {one, one};
              const half2_t p0 = {u[r][0], u[r][1]}, p1 = {u[r][2], u[r][3]};
              csum[0] = __builtin_amdgcn_fdot2(p0, w2, csum[0], false);
              csum[1] = __builtin_amdgcn_fdot2(p1, w2, csum[1], false);
              csq[0] = __builtin_amdgcn_fdot2(p0, p0 * w2, csq[0], false);
              csq[1] = __builtin_amdgcn_fdot2(p1, p1 * w2, csq[1], false);
            }
          }
        }
      };
      // two pieces in flight; every wait names the operations younger than the piece it needs (loads issued after it, stores)
      if (NSRC) { issue(pc0, 0, ReswIC<0>{}); issue(pc1, 0, ReswIC<1>{}); resw_wait<PL, SRC>(pc0); }
      compute(pc0, ReswIC<0>{}, ReswIC<0>{});
      if (NSRC) { issue(pc0, 1, ReswIC<0>{}); resw_wait<PL, SRC>(pc1); }
      compute(pc1, ReswIC<0>{}, ReswIC<1>{});
      if (NSRC) issue(pc1, 1, ReswIC<1>{});
      colsum(0);
#ifdef RESW_X_STAMPS
      unsigned long long qa, qb;
      RSTAMP(qa)
#endif
      rows(0);
#ifdef RESW_X_STAMPS
      RSTAMP(qb)
      b_s += qb - qa; b_e -= qb - qa;
#endif
      if (NSRC) resw_wait<5 + PL, SRC>(pc0);
      compute(pc0, ReswIC<1>{}, ReswIC<0>{});
      if (NSRC) resw_wait<5, SRC>(pc1);
      compute(pc1, ReswIC<1>{}, ReswIC<1>{});
      RSTAMP(q2)
      colsum(1);
      rows(1);
      if (CS && cs_on && lane < 40)
        *(float4_t*)(p.colstats + ((long long)blk * (p.N / 2) + slab * (RW_SLAB / 2) + 2 * lane) * 2) =
            (float4_t){csum[0], csq[0], csum[1], csq[1]};
    } else {
      // slab rows [0,80) = hidden, [80,160) = gate of output columns slab*80 + [0,80)
      auto epi = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int mrow0 = blk * 32 + j * 16;
        float4_t bh[5], bg[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          bh[i] = *(const float4_t*)(bsm + i * 16 + 4 * lq);
          bg[i] = *(const float4_t*)(bsm + 80 + i * 16 + 4 * lq);
        }
        resw_for5([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const float4_t hv = resw_read_acc<(2 * i + j) * 4>() + bh[i];
          const float4_t gv = resw_read_acc<(2 * (i + 5) + j) * 4>() + bg[i];
          const float2_t lo = __builtin_shufflevector(hv, hv, 0, 1) * gelu_erf2(__builtin_shufflevector(gv, gv, 0, 1));
          const float2_t hi = __builtin_shufflevector(hv, hv, 2, 3) * gelu_erf2(__builtin_shufflevector(gv, gv, 2, 3));
          const half4_t o = {(half_t)lo.x, (half_t)lo.y, (half_t)hi.x, (half_t)hi.y};
          *(half4_t*)(scr + l15 * RW_PITCH + (i * 16 + 4 * lq) * 2) = o;
        });
        half8_t v[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const int q = lane + 64 * kk;
          const int row = ((q * 205) >> 11) & 15;     // q / 10 for q < 160 (lanes past the 160 pieces are masked off)
          const int cc = q - ((q * 205) >> 11) * 10;
          v[kk] = *(const half8_t*)(scr + row * RW_PITCH + (cc & 15) * 16);
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          const int q = lane + 64 * kk;
          const int row = ((q * 205) >> 11) & 15;
          const int cc = q - ((q * 205) >> 11) * 10;
          const int mr = q < 160 ? mrow0 + row : 0x7fffffff;
          resw_store16(outp + (unsigned long long)(unsigned)(mrow0 + row) * (unsigned)p.ldc + slab * 80 + cc * 8, v[kk], mr, p.M);
        }
      };
      epi(ReswIC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      RSTAMP(q2)
      epi(ReswIC<1>{});
    }
#ifdef RESW_X_STAMPS
    RSTAMP(q3)
    b_k += q1 - q0; b_e += q2 - q1; b_s += q3 - q2; b_n += 1;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RESW_X_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = lkgd_resw_stamps + (blockIdx.x * 8 + w) * 4;
    o[0] = b_k; o[1] = b_e; o[2] = b_s; o[3] = b_n;
  }
#endif
}

// Which (K, epilogue sources) instantiations exist: the ones hipcc allocates WITHOUT scratch spills inside the 96-register
// budget (tests/test_host_cpu.py compiles this file with -Rpass-analysis and checks it).  A spill would be more than
// slow here: the residual loads are issued and awaited by hand, and spill code placed between a load and its wait would
// save a register whose data has not arrived yet.
static constexpr bool resw_has(int nk, int src) { return nk >= 4 ? src <= 5 : (src == 0 || src == 1 || src == 2 || src == 4 || src == 6); }
static constexpr bool resw_has_cs(int nk, int src) { return src == 0 || src == 2; }      // bare / bias, + one residual (proj_out)

template <int NK>
static void resw_launch_nk(const lkgd_gemm_desc* d, hipStream_t stream, int grid, int lds, int nslab, int per, bool attr_only) {
  const int src = (d->rowbias ? 1 : 0) | (d->res1 ? 2 : 0) | (d->res2 ? 4 : 0);
  const bool cs = d->colstats != nullptr;
#define RW_ONE(G, S, C)                                                                                                 \
  if constexpr (G || (C ? resw_has_cs(NK, S) : resw_has(NK, S))) {                                                      \
    if (attr_only)                                                                                                      \
      (void)hipFuncSetAttribute((const void*)lkgd_gemm_resw_kernel<NK, G, S, C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    else if (G ? d->geglu != 0 : (d->geglu == 0 && src == S && cs == C))                                                \
      hipLaunchKernelGGL((lkgd_gemm_resw_kernel<NK, G, S, C>), dim3(grid), dim3(RW_NT), lds, stream, *d, nslab, per);   \
  }
  RW_ONE(true, 0, false)
  RW_ONE(false, 0, false) RW_ONE(false, 1, false) RW_ONE(false, 2, false) RW_ONE(false, 3, false)
  RW_ONE(false, 4, false) RW_ONE(false, 5, false) RW_ONE(false, 6, false) RW_ONE(false, 7, false)
  RW_ONE(false, 0, true) RW_ONE(false, 2, true)
#undef RW_ONE
}

// may a launch of this descriptor on the resident-weight kernel carry column statistics?
extern "C" int lkgd_gemm_resw_colstats_ok(const lkgd_gemm_desc* d) {
  const int src = (d->rowbias ? 1 : 0) | (d->res1 ? 2 : 0) | (d->res2 ? 4 : 0);
  return !d->geglu && resw_has_cs(d->K / 64, src);
}

// does the resident-weight kernel cover this problem on a device with `cus` compute units?  (lkgd_gemm_f16's dispatch)
extern "C" int lkgd_gemm_resw_ok(const lkgd_gemm_desc* d, int cus) {
  const int nk = d->K / 64;
  if (nk < 1 || nk > 5 || d->K % 64 || d->mode != LKGD_A_PLAIN || d->csplit < d->K) return 0;
  if (d->N % RW_SLAB || (d->geglu != 0 && d->geglu != 80) || d->N / RW_SLAB > cus / 8) return 0;
  if (d->geglu && (d->rowbias || d->res1 || d->res2)) return 0;
  const int src = (d->rowbias ? 1 : 0) | (d->res1 ? 2 : 0) | (d->res2 ? 4 : 0);
  if (!d->geglu && !resw_has(nk, src)) return 0;
  // 16-byte row pieces of out, 8-byte pieces of the sources, 16-byte token fragments; 32-bit row arithmetic
  if (d->ldc % 8 || !aligned16(d->out) || d->lda0 % 8 || !aligned16(d->a0) || !aligned16(d->w)) return 0;
  if ((d->res1 && (d->ldr1 % 4 || ((uintptr_t)d->res1 & 7))) || (d->res2 && (d->ldr2 % 4 || ((uintptr_t)d->res2 & 7))) ||
      (d->rowbias && (d->ldrb % 4 || ((uintptr_t)d->rowbias & 7))))
    return 0;
  return d->M >= 1 && d->M < (1 << 26);
}

// grid = 8 XCD classes x (per x nslab) workgroups: every XCD runs all nslab slabs over its eighth of the rows
extern "C" int lkgd_gemm_resw_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  if (!lkgd_gemm_resw_ok(d, cus)) return LKGD_E_SHAPE;
  if (d->colstats && !lkgd_gemm_resw_colstats_ok(d)) return LKGD_E_SHAPE;
  const int nk = d->K / 64;
  const int nslab = d->N / RW_SLAB;
  const int per = (cus / 8) / nslab;
  const int lds = 10 * nk * 2 * 1024 + RW_SLAB * 4 + RW_WAVES * RW_STAGE;
  const int mx = 10 * 10 * 1024 + RW_SLAB * 4 + RW_WAVES * RW_STAGE;
  const int grid = 8 * per * nslab;
#define RW_NK(ATTR)                                                                                     \
  switch (ATTR ? 0 : nk) {                                                                              \
    case 0:                                                                                             \
    case 1: resw_launch_nk<1>(d, stream, grid, ATTR ? mx : lds, nslab, per, ATTR); if (!ATTR) break;    \
    case 2: resw_launch_nk<2>(d, stream, grid, ATTR ? mx : lds, nslab, per, ATTR); if (!ATTR) break;    \
    case 3: resw_launch_nk<3>(d, stream, grid, ATTR ? mx : lds, nslab, per, ATTR); if (!ATTR) break;    \
    case 4: resw_launch_nk<4>(d, stream, grid, ATTR ? mx : lds, nslab, per, ATTR); if (!ATTR) break;    \
    default: resw_launch_nk<5>(d, stream, grid, ATTR ? mx : lds, nslab, per, ATTR); break;              \
  }
  LKGD_DEVICE_ONCE_BEGIN
    RW_NK(true)
    if (hipGetLastError() != hipSuccess) return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  RW_NK(false)
#undef RW_NK
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
