// 256x256 "ping-pong" MFMA GEMM / implicit-conv kernel (include/lkgd_hip.h section 1) for deep-K, wide-N problems.
//
// Why another tile program: the 256x128 streaming kernel needs 48 KiB of operands per 64-deep K-step for 1024 MFMA cycles
// per SIMD = 48 B/cycle/CU from L2, more than the ~35 B/cycle a CU's LDS-DMA path delivers (MI355X_MICROARCH.md, LDS
// gather from the XCD's L2: 66-73 GB/s per CU), and its 64x64 wave tiles read 1 KiB of LDS per MFMA.  A 256x256 tile
// with 128x64 wave tiles needs 32 B/cycle and 0.75 KiB per MFMA.  The schedule is the guide's "256^2 8-phase" structure
// (cdna_hip_programming.md section 5) re-derived for this kernel's operands:
//
//   * 512 threads = 8 waves as 2 (token groups) x 4 (channel groups); a wave owns 128 tokens x 64 channels, taken as
//     64 + 64 tokens from the two A half-tiles and 32 + 32 channels from the two B half-tiles, so EVERY wave reads every
//     half-tile and a half-tile is completely consumed by one phase;
//   * LDS = 2 K-tile buffers x {A-h0, A-h1, B-h0, B-h1} x 16 KiB, filled by LDS-DMA (2 x global_load_lds_dwordx4 per
//     thread per half-tile) from a linear stream of half-tiles that runs across K-tiles AND output tiles;
//   * a K-tile is 4 phases of 8 x v_mfma_f32_32x32x16_f16 (one 64x32 quadrant x K=64 each):
//        P1: read B-h0 (4 x ds_read_b128), A-h0 (8)   stage A-h1 of K-tile s+1     MFMA (A0,B0)
//        P2: read B-h1 (4)                            stage B-h0 of K-tile s+2     MFMA (A0,B1)
//        P3: read A-h1 (8)                            stage A-h0 of K-tile s+2     MFMA (A1,B1)
//        P4: -                                        stage B-h1 of K-tile s+2     MFMA (A1,B0)
//     P4, P1 and P2 each end their read section with s_waitcnt vmcnt(10): the half-tile(s) the NEXT phase reads have
//     landed, the five newest half-tiles stay in flight (80 KiB per CU; every half-tile gets >= 5 phases to arrive);
//     every phase is  { reads ; stage ; s_barrier ; lgkmcnt(0) ; MFMAs ; s_barrier }.  A half-tile is restaged at the
//     earliest two phases after the phase that read it (one phase after for B-h0, whose reads are retired by an
//     lgkmcnt before that phase's first barrier); it is read at the earliest one phase after the counted vmcnt + barrier
//     that retired it;
//   * the two token groups (waves 0-3 / 4-7; wave w and w+4 share a SIMD) run ONE BARRIER APART: while one group's
//     MFMA cluster owns the matrix pipe the other group reads LDS and issues DMA.  They re-align for the epilogue;
//   * registers: 128 accumulators + 64 fragment registers leave ~60 for everything else, and hipcc's allocator does not
//     get there from C++ (it keeps the two A fragment sets apart and spills into the K-loop, where every scratch reload
//     drains vmcnt).  The K-tile body is therefore ONE inline-asm block: accumulators live in a[0:127] (named in the asm
//     text, declared as clobbers), fragments and address temporaries are scratch operands, the eight source addresses of
//     the K-tile's DMA loads are inputs computed by the C++ around it;
//   * operands swapped in the MFMA (A = weight rows, B = token rows) as in gemm_stream.hip: lane = token, registers =
//     4 consecutive channels; the epilogue moves one 32-token fragment at a time out of the AGPRs and transposes through a
//     private 4 KiB LDS scratch per wave to whole 128-byte rows.
#include "gemm_common.h"

#define PBM 256
#define PBN 256
#define PNT 512
#define PHALF 16384                      // 128 rows x 64 k x 2 B
#define PSCR_OFF (8 * PHALF)             // 128 KiB of tiles, then 8 x 4 KiB epilogue scratch
#define PLDS (PSCR_OFF + 8 * 4096)       // 160 KiB

__device__ __forceinline__ float gelu_fast_pp(float x) {
  // exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below fp16 resolution)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_s = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erf_s);
}

// ---- staging state: the lean gather of gemm_common.h (four A rows per thread) + the weight row and stream position
struct PPStage : LeanGather<4> {
  int n0;                       // channel of LDS row srow of B half 0
  int tile, kt, s;              // tile / K-tile / stream index of the K-tile the state describes       (wave-uniform)
};

#define PP_GW 4     // supertile width (gemm_common.h)

// move the staging state to the next K-tile of the stream (stays on the last one at the end of the stream: the extra
// loads the uniform schedule issues there re-read valid memory into buffers nobody reads any more)
template <int MODE>
__device__ __forceinline__ void pp_advance(const lkgd_gemm_desc& p, PPStage& st, int total, int nk, int nc, int tiles_m,
                                           int tiles_n,
                                           int srow, int schunk, float rcp0, float rcp1) {
  if (st.s + 1 >= total) return;
  ++st.s;
  if (++st.kt == nk) {
    st.kt = 0;
    st.tile += nc;
    int tm, tn;
    supertile<PP_GW>(st.tile, tiles_m, tiles_n, tm, tn);
#pragma unroll
    for (int i = 0; i < 4; ++i) st.rd[i] = lean_row<MODE>(p, tm * PBM + srow + 64 * i, rcp0, rcp1);
    // LDS row rr of B half hb holds channel  tn*256 + (rr>>5)*64 + hb*32 + (rr&31): a wave's 32 + 32 channels are then
    // 64 CONSECUTIVE output channels.  This thread fills rows srow and srow+64 (= +128 channels) of both halves.
    st.n0 = tn * PBN + (srow >> 5) * 64 + (srow & 31);
    st.seg_end = 0;
  }
  if (st.kt * BK >= st.seg_end) lean_segment<MODE, 4>(p, st, st.kt * BK, schunk);
}

__device__ __forceinline__ const half_t* pp_a(const PPStage& st, int i) {
  return st.aptr[i] + (((st.zmask >> i) & 1u) ? 0 : (st.kt * BK - st.seg_k0));
}
// byte offset of weight row n0 + dn (chunk included) from  weights + K offset
__device__ __forceinline__ unsigned pp_boff(const lkgd_gemm_desc& p, const PPStage& st, int dn, int schunk) {
  int n = st.n0 + dn;
  n = n < p.N ? n : p.N - 1;      // clamped: channels past N are computed on a copy of the last row and never stored
  return ((unsigned)n * (unsigned)p.K + schunk * 8) * 2u;
}
__device__ __forceinline__ const half_t* pp_b(const lkgd_gemm_desc& p, const PPStage& st, int dn, int schunk) {
  return (const half_t*)((const char*)p.w + pp_boff(p, st, dn, schunk)) + st.kt * BK;
}

// ---- the K-tile body ------------------------------------------------------------------------------------------------
// acc[i][j] = a[(4i + j)*16 .. +15]: channel half i, token fragment j = 2*(A half) + (32-row fragment of the wave's 64).
// Register budget per wave: 128 AGPRs (accumulators) + 128 VGPRs: 64 fragments, 3 address temporaries, 3 read bases,
// 8 (A source addresses) + 4 (B source offsets), the rest is the C++ staging state.
// timing experiments (tools/micro/pp_knobs.sh): drop one ingredient of the K-tile body; results are then wrong
#ifdef PP_X_NOREAD
#define PP_RD(X) ""
#else
#define PP_RD(X) X
#endif
#ifdef PP_X_NOSTAGE
#define PP_LD(X) ""
#else
#define PP_LD(X) X
#endif
#ifdef PP_X_NOBAR
#define PP_BAR ""
#else
#define PP_BAR "s_barrier\n\t"
#endif
#ifdef PP_X_NOMFMA
#define PP_MM(X) ""
#else
#define PP_MM(X) X
#endif
#define PP_XOR3(BASE) /* %16..%18 = BASE ^ 32, 64, 96: the K-steps of one fragment row */       \
  "v_xor_b32 %16, 32, " BASE "\n\t"                                                             \
  "v_xor_b32 %17, 64, " BASE "\n\t"                                                             \
  "v_xor_b32 %18, 96, " BASE "\n\t"
#define PP_DS4(D0, D1, D2, D3, BASE, OFF)                                                       \
  PP_RD("ds_read_b128 " D0 ", " BASE " offset:" OFF "\n\t")                                     \
  PP_RD("ds_read_b128 " D1 ", %16 offset:" OFF "\n\t")                                          \
  PP_RD("ds_read_b128 " D2 ", %17 offset:" OFF "\n\t")                                          \
  PP_RD("ds_read_b128 " D3 ", %18 offset:" OFF "\n\t")
#define PP_STAGE_A(MREG, SLOT, P0, P1)                                                          \
  "s_add_u32 m0, " MREG ", " SLOT "\n\t"                                                        \
  "s_nop 0\n\t"                                                                                 \
  PP_LD("global_load_lds_dwordx4 " P0 ", off\n\t")                                              \
  "s_add_u32 m0, m0, 8192\n\t"                                                                  \
  "s_nop 0\n\t"                                                                                 \
  PP_LD("global_load_lds_dwordx4 " P1 ", off\n\t")
#define PP_STAGE_B(MREG, SLOT, O0, O1, WK)                                                      \
  "s_add_u32 m0, " MREG ", " SLOT "\n\t"                                                        \
  "s_nop 0\n\t"                                                                                 \
  PP_LD("global_load_lds_dwordx4 " O0 ", " WK "\n\t")                                           \
  "s_add_u32 m0, m0, 8192\n\t"                                                                  \
  "s_nop 0\n\t"                                                                                 \
  PP_LD("global_load_lds_dwordx4 " O1 ", " WK "\n\t")
// 8 MFMAs of one quadrant: acc AC0 (token fragment jj = 0) and AC1 (jj = 1) += B[ks] x F[jj][ks]
#define PP_MFMA8(AC0, AC1, C0, C1, B0, B1, B2, B3)                                              \
  "s_setprio 1\n\t"                                                                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC0 ", " B0 ", %0, " C0 "\n\t")                              \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC1 ", " B0 ", %4, " C1 "\n\t")                              \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC0 ", " B1 ", %1, " AC0 "\n\t")                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC1 ", " B1 ", %5, " AC1 "\n\t")                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC0 ", " B2 ", %2, " AC0 "\n\t")                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC1 ", " B2 ", %6, " AC1 "\n\t")                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC0 ", " B3 ", %3, " AC0 "\n\t")                             \
  PP_MM("v_mfma_f32_32x32x16_f16 " AC1 ", " B3 ", %7, " AC1 "\n\t")                             \
  "s_setprio 0\n\t"

#define PP_AGPR_CLOBBERS                                                                                             \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",  \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33",    \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49",    \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65",    \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81",    \
  "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97",    \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111",      \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125",    \
  "a126", "a127"

// The K-tile body is two asm statements - P1..P3, then P4 - so that the C++ that prepares the NEXT K-tile's DMA sources
// runs between them, inside P4's short read section (P4 reads nothing from LDS), instead of between two K-tiles where it
// would sit in front of P1's twelve reads on the critical path of both wave groups.
// FIRST: the K-tile opens an output tile - its MFMAs take 0 as the accumulator input instead of a[]
struct PPIn {                   // DMA sources of one K-tile body
  const half_t* pA2; const half_t* pA3;     // A-h1 rows of K-tile s+1      (P1)
  const half_t* pA0; const half_t* pA1;     // A-h0 rows of K-tile s+2      (P3)
  unsigned oB0, oB1, oB2, oB3;              // byte offsets of the weight rows of K-tile s+2 from wk (P2: 0,1; P4: 2,3)
  const half_t* wk;                          // weights + K offset of K-tile s+2 (wave-uniform)
};

template <bool FIRST>
__device__ __forceinline__ void pp_ktile_a(half8_t (&f)[8], half8_t (&g)[4], int xb0, int xb1, int wb, const PPIn& in,
                                           int m_other, int m_this) {
  half8_t h0, h1, h2, h3;
  int t0, t1, t2;
  // operands: %0-%7 F[jj][ks] (jj*4+ks), %8-%11 B0[ks], %12-%15 B1[ks], %16-%18 address temporaries,
  //           %19 xb0, %20 xb1 (A read bases of token fragment jj), %21 wb (B read base), %22 pA2, %23 pA3,
  //           %24 pA0, %25 pA1, %26 oB0, %27 oB1, %28 wk (SGPR pair),
  //           %29 m_other (LDS base of the other K-tile buffer + this wave's 1 KiB slice), %30 m_this
#define PP_BODY_A(C00, C01, C10, C11, C12, C13)                                                                      \
  asm volatile(                                                                                                      \
      "s_waitcnt lgkmcnt(0)\n\t" /* no scalar load of the surrounding code may sit in the counted LDS waits */       \
      /* ---------------------------------------------------------------------------------------- P1 */            \
      PP_XOR3("%21") PP_DS4("%8", "%9", "%10", "%11", "%21", "0")                                                    \
      PP_XOR3("%19") PP_DS4("%0", "%1", "%2", "%3", "%19", "0")                                                      \
      PP_XOR3("%20") PP_DS4("%4", "%5", "%6", "%7", "%20", "0")                                                      \
      PP_STAGE_A("%29", "16384", "%22", "%23")                                                                       \
      "s_waitcnt vmcnt(10)\n\t" /* B-h1 of this K-tile has landed (P2 reads it): five newer half-tiles in flight */  \
      "s_waitcnt lgkmcnt(8)\n\t"                                                                                     \
      PP_BAR                                                                                                         \
      "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
      PP_MFMA8("a[0:15]", "a[16:31]", C00, C01, "%8", "%9", "%10", "%11")                                            \
      PP_BAR /* ---------------------------------------------------------------------------------- P2 */            \
      PP_XOR3("%21") PP_DS4("%12", "%13", "%14", "%15", "%21", "16384")                                              \
      PP_STAGE_B("%30", "32768", "%26", "%27", "%28")                                                                \
      "s_waitcnt vmcnt(10)\n\t" /* A-h1 of this K-tile has landed (P3 reads it) */                                   \
      PP_BAR                                                                                                         \
      "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
      PP_MFMA8("a[64:79]", "a[80:95]", C10, C11, "%12", "%13", "%14", "%15")                                         \
      PP_BAR /* ---------------------------------------------------------------------------------- P3 */            \
      PP_XOR3("%19") PP_DS4("%0", "%1", "%2", "%3", "%19", "16384")                                                  \
      PP_XOR3("%20") PP_DS4("%4", "%5", "%6", "%7", "%20", "16384")                                                  \
      PP_STAGE_A("%30", "0", "%24", "%25")                                                                           \
      PP_BAR                                                                                                         \
      "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
      PP_MFMA8("a[96:111]", "a[112:127]", C12, C13, "%12", "%13", "%14", "%15")                                      \
      PP_BAR                                                                                                         \
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]),     \
        "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(t0),  \
        "=&v"(t1), "=&v"(t2)                                                                                         \
      : "v"(xb0), "v"(xb1), "v"(wb), "v"(in.pA2), "v"(in.pA3), "v"(in.pA0), "v"(in.pA1), "v"(in.oB0), "v"(in.oB1),  \
        "s"(in.wk), "s"(m_other), "s"(m_this)                                                                        \
      : "memory", "scc", PP_AGPR_CLOBBERS)
  if (FIRST) {
    PP_BODY_A("0", "0", "0", "0", "0", "0");
  } else {
    PP_BODY_A("a[0:15]", "a[16:31]", "a[64:79]", "a[80:95]", "a[96:111]", "a[112:127]");
  }
#undef PP_BODY_A
}

template <bool FIRST>
__device__ __forceinline__ void pp_ktile_b(const half8_t (&f)[8], const half8_t (&g)[4], unsigned oB2, unsigned oB3,
                                           const half_t* wk, int m_this) {
  // operands: %0-%7 F (A-h1 fragments read in P3), %8-%11 B0 (read in P1), %12 oB2, %13 oB3, %14 wk, %15 m_this
#define PP_BODY_B(C02, C03)                                                                                          \
  asm volatile(                                                                                                      \
      "s_add_u32 m0, %15, 49152\n\t"                                                                                 \
      "s_nop 0\n\t"                                                                                                  \
      PP_LD("global_load_lds_dwordx4 %12, %14\n\t")                                                                  \
      "s_add_u32 m0, m0, 8192\n\t"                                                                                   \
      "s_nop 0\n\t"                                                                                                  \
      PP_LD("global_load_lds_dwordx4 %13, %14\n\t")                                                                  \
      "s_waitcnt vmcnt(10)\n\t" /* B-h0 and A-h0 of the next K-tile have landed (its P1 reads them) */              \
      PP_BAR                                                                                                         \
      PP_MFMA8("a[32:47]", "a[48:63]", C02, C03, "%8", "%9", "%10", "%11")                                           \
      PP_BAR                                                                                                         \
      :                                                                                                              \
      : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(g[0]),          \
        "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(oB2), "v"(oB3), "s"(wk), "s"(m_this)                                    \
      : "memory", "scc", PP_AGPR_CLOBBERS)
  if (FIRST) {
    PP_BODY_B("0", "0");
  } else {
    PP_BODY_B("a[32:47]", "a[48:63]");
  }
#undef PP_BODY_B
}

// one accumulator fragment out of the AGPRs (BASE is a literal register index)
template <int BASE>
__device__ __forceinline__ float16_t pp_read_acc() {
  float16_t v;
  // one statement, "memory": the reads of fragment j+1 stay behind the stores of fragment j (register pressure)
  float x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%16]\n\tv_accvgpr_read_b32 %1, a[%16+1]\n\t"
      "v_accvgpr_read_b32 %2, a[%16+2]\n\tv_accvgpr_read_b32 %3, a[%16+3]\n\t"
      "v_accvgpr_read_b32 %4, a[%16+4]\n\tv_accvgpr_read_b32 %5, a[%16+5]\n\t"
      "v_accvgpr_read_b32 %6, a[%16+6]\n\tv_accvgpr_read_b32 %7, a[%16+7]\n\t"
      "v_accvgpr_read_b32 %8, a[%16+8]\n\tv_accvgpr_read_b32 %9, a[%16+9]\n\t"
      "v_accvgpr_read_b32 %10, a[%16+10]\n\tv_accvgpr_read_b32 %11, a[%16+11]\n\t"
      "v_accvgpr_read_b32 %12, a[%16+12]\n\tv_accvgpr_read_b32 %13, a[%16+13]\n\t"
      "v_accvgpr_read_b32 %14, a[%16+14]\n\tv_accvgpr_read_b32 %15, a[%16+15]"
      : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=v"(x8), "=v"(x9),
        "=v"(x10), "=v"(x11), "=v"(x12), "=v"(x13), "=v"(x14), "=v"(x15)
      : "i"(BASE)
      : "memory");
  v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6; v[7] = x7;
  v[8] = x8; v[9] = x9; v[10] = x10; v[11] = x11; v[12] = x12; v[13] = x13; v[14] = x14; v[15] = x15;
  return v;
}

template <int MODE>
__global__ __launch_bounds__(PNT) void lkgd_gemm_pp_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int h = lane >> 5, l31 = lane & 31;

  // ---- tile schedule (as gemm_stream.hip): each XCD label owns a contiguous tile range, its workgroups take it round-robin
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  const int nc = (G - xcd + 7) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);
  const int xe = xb + q8 + (xcd < r8 ? 1 : 0);
  const int my_tiles = (xe - xb - c + nc - 1) / nc;
  const int tile_begin = xb + c;
  const int nk = p.K / BK;
  if (my_tiles <= 0) return;
  const int total = my_tiles * nk;          // K-tiles in this workgroup's stream

  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  float rcp0, rcp1;                         // reciprocals of the row-index divisors (uniform)
  lean_rcps<MODE>(p, rcp0, rcp1);
  PPStage st;
  st.tile = tile_begin - nc; st.kt = nk - 1; st.s = -1; st.seg_k0 = 0; st.seg_end = 0; st.zmask = 0; st.n0 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { st.rd[i].base = -1; st.rd[i].yx = 0; st.aptr[i] = (const half_t*)p.zeros; }
#define ADVANCE() pp_advance<MODE>(p, st, total, nk, nc, tiles_m, tiles_n, srow, schunk, rcp0, rcp1)

  // LDS read bases inside a K-tile buffer.  Token rows of this wave in an A half: wr*64 + jj*32 + l31; channel rows in a
  // B half: wc*32 + l31.  Chunk swizzle (row>>1)&7 as written by the staging side:  chunk (2ks + h) ^ sw = (2ks) ^ (h ^ sw),
  // so the K-step is one XOR with ks << 5 on a per-lane base.
  const int rx0 = wr * 64 + l31, rx1 = rx0 + 32, rw = wc * 32 + l31;
  const int x_base0 = rx0 * 128 + ((h ^ ((rx0 >> 1) & 7)) << 4);
  const int x_base1 = rx1 * 128 + ((h ^ ((rx1 >> 1) & 7)) << 4);
  const int w_base = 2 * PHALF + rw * 128 + ((h ^ ((rw >> 1) & 7)) << 4);

  // ---- pipeline fill: K-tile 0 complete + B-h0, A-h0, B-h1 of K-tile 1 in flight
  {
    ADVANCE();                                           // K-tile 0 -> buffer 0
    char* b0 = smem + w * 1024;
    glds16(pp_b(p, st, 0, schunk), b0 + 2 * PHALF);          glds16(pp_b(p, st, 128, schunk), b0 + 2 * PHALF + 8192);
    glds16(pp_a(st, 0), b0);                                  glds16(pp_a(st, 1), b0 + 8192);
    glds16(pp_b(p, st, 32, schunk), b0 + 3 * PHALF);         glds16(pp_b(p, st, 160, schunk), b0 + 3 * PHALF + 8192);
    glds16(pp_a(st, 2), b0 + PHALF);                          glds16(pp_a(st, 3), b0 + PHALF + 8192);
    ADVANCE();                                           // K-tile 1 -> buffer 1 (re-reads K-tile 0 when total == 1)
    char* b1 = smem + 4 * PHALF + w * 1024;
    glds16(pp_b(p, st, 0, schunk), b1 + 2 * PHALF);          glds16(pp_b(p, st, 128, schunk), b1 + 2 * PHALF + 8192);
    glds16(pp_a(st, 0), b1);                                  glds16(pp_a(st, 1), b1 + 8192);
    glds16(pp_b(p, st, 32, schunk), b1 + 3 * PHALF);         glds16(pp_b(p, st, 160, schunk), b1 + 3 * PHALF + 8192);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const half_t* wbase = (const half_t*)p.w;
  asm volatile("" : "+s"(wbase));            // an opaque SGPR pair: kept (or spilled to a lane), never re-loaded from kernarg
  // DMA sources of a K-tile body: A-h1 of K-tile s+1 (the state), then B-h0, A-h0, B-h1 of K-tile s+2
  auto next_in = [&]() {
    PPIn in;
    in.pA2 = pp_a(st, 2);
    in.pA3 = pp_a(st, 3);
    ADVANCE();
    in.pA0 = pp_a(st, 0);
    in.pA1 = pp_a(st, 1);
    in.oB0 = pp_boff(p, st, 0, schunk);  in.oB1 = pp_boff(p, st, 128, schunk);
    in.oB2 = pp_boff(p, st, 32, schunk); in.oB3 = pp_boff(p, st, 160, schunk);
    in.wk = wbase + st.kt * BK;
    return in;
  };
  PPIn in = next_in();

  int kt = 0, tile = tile_begin;
  half8_t f[8], g[4];
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    if (kt == 0 && wr == 1) __builtin_amdgcn_s_barrier();  // token group 1 runs one barrier behind group 0
    const int boff = (s & 1) * (4 * PHALF);
    const int m_this = boff + w * 1024, m_other = (boff ^ (4 * PHALF)) + w * 1024;
    if (kt == 0) pp_ktile_a<true>(f, g, x_base0 + boff, x_base1 + boff, w_base + boff, in, m_other, m_this);
    else pp_ktile_a<false>(f, g, x_base0 + boff, x_base1 + boff, w_base + boff, in, m_other, m_this);
    // P4's read section: the next K-tile's sources are prepared here
    const unsigned oB2 = in.oB2, oB3 = in.oB3;
    const half_t* wk4 = in.wk;
    in = next_in();
    if (kt == 0) pp_ktile_b<true>(f, g, oB2, oB3, wk4, m_this);
    else pp_ktile_b<false>(f, g, oB2, oB3, wk4, m_this);

    if (++kt < nk) continue;
    kt = 0;
    if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the token groups: both run the epilogue together
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results must have left the matrix pipe
    // -------------------------------------------------------------------- epilogue of `tile`, straight from registers
    int tm, tn;
    supertile<PP_GW>(tile, tiles_m, tiles_n, tm, tn);
    tile += nc;
    char* scr = smem + PSCR_OFF + w * 4096;              // 32 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7)
    const int n0 = tn * PBN + wc * 64;
    const half_t* rbp = (const half_t*)p.rowbias;
    const half_t* r1p = (const half_t*)p.res1;
    const half_t* r2p = (const half_t*)p.res2;
    half_t* outp = (half_t*)p.out;
    const int crow = lane >> 3, cchunk = lane & 7;       // coalesced map: 8 lanes per 128-byte row, 8 rows per pass
    const int sw31 = l31 & 7;
    auto epilogue = [&](int j, float16_t e0, float16_t e1) {
      float16_t e[2] = {e0, e1};
      const long long mb = (long long)tm * PBM + (j >> 1) * 128 + wr * 64 + (j & 1) * 32;
      if (!p.geglu) {
        const int ncol = n0 + cchunk * 8;
        uint4 rres[4];
        if (r1p) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long mr = mb + crow + 8 * it;
            rres[it] = (mr < p.M && ncol < p.N) ? *(const uint4*)(r1p + mr * p.ldr1 + ncol) : uint4{0u, 0u, 0u, 0u};
          }
        }
        const long long m = mb + l31;
        long long idx = 0;
        if (rbp && m < p.M) idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) + (unsigned)p.rb_c0) % (unsigned)p.rb_md;   // M is an int32
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = n0 + i * 32 + 8 * g + 4 * h;
            float4_t v;
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = e[i][4 * g + x];
            if (n < p.N) {
              if (p.bias) v += *(const float4_t*)(p.bias + n);
              if (rbp && m < p.M) {
                half4_t rb = *(const half4_t*)(rbp + idx * p.ldrb + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += (float)rb[x];
              }
            }
            v *= p.s_acc;
#pragma unroll
            for (int x = 0; x < 4; ++x) e[i][4 * g + x] = v[x];
          }
        if (r1p) {
#pragma unroll
          for (int it = 0; it < 4; ++it)
            *(uint4*)(scr + (crow + 8 * it) * 128 + ((cchunk ^ crow) << 4)) = rres[it];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              half4_t r = *(const half4_t*)(scr + l31 * 128 + (((i * 4 + g) ^ sw31) << 4) + 8 * h);
#pragma unroll
              for (int x = 0; x < 4; ++x) e[i][4 * g + x] += p.r1 * (float)r[x];
            }
        }
        if (r2p) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long mr = mb + crow + 8 * it;
            uint4 x = {0u, 0u, 0u, 0u};
            if (mr < p.M && ncol < p.N) x = *(const uint4*)(r2p + mr * p.ldr2 + ncol);
            *(uint4*)(scr + (crow + 8 * it) * 128 + ((cchunk ^ crow) << 4)) = x;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              half4_t r = *(const half4_t*)(scr + l31 * 128 + (((i * 4 + g) ^ sw31) << 4) + 8 * h);
#pragma unroll
              for (int x = 0; x < 4; ++x) e[i][4 * g + x] += p.r2 * (float)r[x];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4_t o;
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] = (half_t)e[i][4 * g + x];
            *(half4_t*)(scr + l31 * 128 + (((i * 4 + g) ^ sw31) << 4) + 8 * h) = o;
          }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const long long mr = mb + crow + 8 * it;
          const uint4 x = *(const uint4*)(scr + (crow + 8 * it) * 128 + ((cchunk ^ crow) << 4));
          if (mr < p.M && ncol < p.N) *(uint4*)(outp + mr * p.ldc + ncol) = x;
        }
      } else {
        // wave channels [0,32) = hidden, [32,64) = gate of output columns tn*128 + wc*32 + [0,32)
        const int oc0 = tn * (PBN / 2) + wc * 32;
        const int grow = lane >> 2, gchunk = lane & 3;     // 4 lanes per 64-byte row, 16 rows per pass
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cc = 8 * g + 4 * h;
          float4_t bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
          if (p.bias && n0 < p.N) {
            bh = *(const float4_t*)(p.bias + n0 + cc);
            bg = *(const float4_t*)(p.bias + n0 + 32 + cc);
          }
          half4_t o;
#pragma unroll
          for (int x = 0; x < 4; ++x)
            o[x] = (half_t)((e[0][4 * g + x] + bh[x]) * gelu_fast_pp(e[1][4 * g + x] + bg[x]));
          *(half4_t*)(scr + l31 * 128 + ((g ^ sw31) << 4) + 8 * h) = o;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const long long mr = mb + grow + 16 * it;
          const uint4 x = *(const uint4*)(scr + (grow + 16 * it) * 128 + ((gchunk ^ (grow & 7)) << 4));
          if (mr < p.M && n0 < p.N) *(uint4*)(outp + mr * p.ldc + oc0 + gchunk * 8) = x;
        }
      }
    };
    epilogue(0, pp_read_acc<0>(), pp_read_acc<64>());
    epilogue(1, pp_read_acc<16>(), pp_read_acc<80>());
    epilogue(2, pp_read_acc<32>(), pp_read_acc<96>());
    epilogue(3, pp_read_acc<48>(), pp_read_acc<112>());
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's trailing loads must land before the LDS is released
}

extern "C" int lkgd_gemm_pp_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lkgd_gemm_pp_kernel<LKGD_A_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            PLDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_pp_kernel<LKGD_A_CONV3X3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            PLDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_pp_kernel<LKGD_A_TCONV3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            PLDS) != hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  if (d->M >= (1 << 24)) return LKGD_E_SHAPE;            // float-reciprocal row decomposition
  int tiles_m = (d->M + PBM - 1) / PBM, tiles_n = (d->N + PBN - 1) / PBN;
  long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  if (d->mode == LKGD_A_PLAIN)
    hipLaunchKernelGGL(lkgd_gemm_pp_kernel<LKGD_A_PLAIN>, dim3(grid), dim3(PNT), PLDS, stream, *d, tiles_m, tiles_n);
  else if (d->mode == LKGD_A_CONV3X3)
    hipLaunchKernelGGL(lkgd_gemm_pp_kernel<LKGD_A_CONV3X3>, dim3(grid), dim3(PNT), PLDS, stream, *d, tiles_m, tiles_n);
  else if (d->mode == LKGD_A_TCONV3)
    hipLaunchKernelGGL(lkgd_gemm_pp_kernel<LKGD_A_TCONV3>, dim3(grid), dim3(PNT), PLDS, stream, *d, tiles_m, tiles_n);
  else
    return LKGD_E_MODE;
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
