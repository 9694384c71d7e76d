// Wide-tile persistent MFMA GEMM / implicit-conv kernel: 256 (tokens) x 320 (channels) output tiles.
//
// Why 320: every channel count on the SVD hot path (320, 640, 960, 1280, 1920, 2560, 3840, 5120, 10240) is a multiple of
// 320, so there is no padded-column waste (128-wide tiles lose 17 % at N = 320), and a 256x320 tile moves
// (256+320)*64*2 B per 10.5 MFLOP K-step = 29 B per MFMA-cycle per CU - within what a CU can pull from its XCD's L2,
// where the 256x128 tile of gemm_stream.hip needs 47 B/cycle and is load-bound.
//
//   * 512 threads, 8 waves as 4(M) x 2(N); a wave owns 64 tokens x 160 channels = 4 x 10 fragments of
//     v_mfma_f32_16x16x32_f16 (160 accumulator VGPRs); BK = 64; two LDS stages of 72 KiB (one K-tile, 2560 MFMA-cycles
//     of cover, always in flight);
//   * persistent workgroups, K-tile stream continuous across output tiles, XCD-cooperative tile schedule and
//     swapped-operand register epilogue exactly as gemm_stream.hip;
//   * GEGLU: packed rows interleave 80 hidden | 80 gate per wave (5 + 5 fragments): both factors in the same lane/register.
#include "gemm_common.h"

// timing-experiment knobs (tools/micro/wide_knobs.sh; results are wrong with NOSTORE): never defined in the product build
#ifdef WIDE_X_NOSTORE
#define WIDE_STORE_GUARD if (p.ldc < 0)
#elif defined(WIDE_X_ONELANE)   /* every store instruction is issued, one lane of it active: the epilogue's arithmetic without its bytes */
#define WIDE_STORE_GUARD if ((threadIdx.x & 63) == 0)
#else
#define WIDE_STORE_GUARD
#endif
#ifdef WIDE_X_SMALLOUT          /* every store lands in the first 2048 rows: the output stays L2-resident */
#define WIDE_OUT_ROW(m) ((m) & 2047)
#else
#define WIDE_OUT_ROW(m) (m)
#endif
#ifdef WIDE_X_SMALLA            /* the A operand comes from the first 4096 rows only: it stays L2-resident */
#define WIDE_A_ROW(m) ((m) & 4095)
#else
#define WIDE_A_ROW(m) (m)
#endif
#ifdef WIDE_X_NT
#define WIDE_ST(ptr, v) __builtin_nontemporal_store(v, ptr)
#else
#define WIDE_ST(ptr, v) (*(ptr) = (v))
#endif
#define WIDE_OUT_PITCH 336      // bytes per staged output row (320 + 16: rows start on different banks)
#define WBM 256
#define WBN 320
#define WNT 512
#define WSTAGE_BYTES ((WBM + WBN) * BK * 2)   // 72 KiB
#define WBIAS_OFF (2 * WSTAGE_BYTES)          // two 320-float bias strips (tile parity) behind the stages
#define WRB_OFF (WBIAS_OFF + 2 * WBN * 4)      // row-bias strips: [tile parity][first | last row's table row][384 halfs]
#define WRB_STRIP 768
#define WLDS (WRB_OFF + 4 * WRB_STRIP)        // 149.5 KiB

// ---- K-tile body (generated asm, tools/gen_wide_asm.py).  hipcc's allocator spills a few values of the C++ form of this
// loop and reloads them next to the LDS-DMA issue; every scratch reload is followed by s_waitcnt vmcnt(0), which in the
// 3x3-conv instantiation sat right behind the nine freshly issued DMA loads and serialised their latency with the MFMAs.
// In the asm form the 160 accumulators are pinned in a[0:159], the fragments are 40 scratch VGPRs, and the only waits are
// the counted lgkmcnt of the software-pipelined fragment reads.
#define WIDE_WAIT_TOP "s_waitcnt lgkmcnt(0)\n\t"   /* no scalar load of the surrounding code in the counted LDS waits */
#ifdef WIDE_X_NOSTAGE
#define WIDE_LD(X) ""
#else
#define WIDE_LD(X) X
#endif
#ifdef WIDE_X_NOREAD
#define WIDE_RD(X) ""
#else
#define WIDE_RD(X) X
#endif
#ifdef WIDE_X_NOMFMA
#define WIDE_MM(X) ""
#else
#define WIDE_MM(X) X
#endif
#if defined(WIDE_X_NOBAR) || defined(WIDE_X_STAMPS)      /* STAMPS: the barrier is issued (and timed) from C++ */
#define WIDE_BAR(X) ""
#else
#define WIDE_BAR(X) X
#endif
#include "gemm_wide_ktile.inc"

struct WideIn {                 // what one K-tile body needs about the NEXT K-tile (its nine DMA loads), besides the
  unsigned oB[5];               // A row sources in LeanGather::aptr: byte offsets of the five weight row chunks from wk
  const half_t* wk;             // weights + K offset (wave-uniform)
};

// statement A: barrier, K-step 0 (fragment reads, the nine DMA loads of the next K-tile, 40 MFMAs) and the reads of K-step 1's
// first fragments, which leave in x1 / w0
template <bool FIRST, int NF, int MF>
__device__ __forceinline__ void wide_ktile_a(half8_t (&x1)[4], half8_t& w0, int xa0, int xa1, int wa0, int wa1,
                                             const half_t* const (&pA)[4], const WideIn& in, int m_a) {
  half8_t x0, x1_, x2, x3, w1;
#define WIDE_STMT(BODY)                                                                                                \
  asm volatile(BODY                                                                                                    \
               : "=&v"(x0), "=&v"(x1_), "=&v"(x2), "=&v"(x3), "=&v"(x1[0]), "=&v"(x1[1]), "=&v"(x1[2]), "=&v"(x1[3]),  \
                 "=&v"(w0), "=&v"(w1)                                                                                  \
               : "v"(xa0), "v"(xa1), "v"(wa0), "v"(wa1), "v"(pA[0]), "v"(pA[1]), "v"(pA[2]), "v"(pA[3]),               \
                 "v"(in.oB[0]), "v"(in.oB[1]), "v"(in.oB[2]), "v"(in.oB[3]), "v"(in.oB[4]), "s"(in.wk), "s"(m_a)       \
               : "memory", "scc", WIDE_AGPR_CLOBBERS)
  if (MF == 3) {              // the 192-row forms: three token fragments per wave, three token loads per thread (x0[3], x1[3], pA[3] unused)
    if (NF == 8) {
      if (FIRST) {
        WIDE_STMT(WIDE8_M3_KTILE_ASM_FIRST_A);
      } else {
        WIDE_STMT(WIDE8_M3_KTILE_ASM_NEXT_A);
      }
    } else if (FIRST) {
      WIDE_STMT(WIDE_M3_KTILE_ASM_FIRST_A);
    } else {
      WIDE_STMT(WIDE_M3_KTILE_ASM_NEXT_A);
    }
  } else if (NF == 8) {       // the 256x256 form: eight weight fragments per wave, four weight loads per thread (in.oB[4] unused)
    if (FIRST) {
      WIDE_STMT(WIDE8_KTILE_ASM_FIRST_A);
    } else {
      WIDE_STMT(WIDE8_KTILE_ASM_NEXT_A);
    }
  } else if (FIRST) {
    WIDE_STMT(WIDE_KTILE_ASM_FIRST_A);
  } else {
    WIDE_STMT(WIDE_KTILE_ASM_NEXT_A);
  }
#undef WIDE_STMT
}

// statement B: K-step 1 (40 MFMAs).  The C++ between A and B prepares the NEXT K-tile's sources while K-step 0's MFMAs drain
// (between two K-tiles it would sit behind the barrier with the matrix pipe idle).
template <int NF, int MF>
__device__ __forceinline__ void wide_ktile_b(const half8_t (&x1)[4], half8_t w0, int wa1) {
  half8_t w1;
#define WIDE_STMT_B(BODY)                                                  \
  asm volatile(BODY                                                        \
               : "=&v"(w1), "+v"(w0)                                       \
               : "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(wa1)  \
               : "memory", WIDE_AGPR_CLOBBERS)
  if (MF == 3) {
    if (NF == 8) WIDE_STMT_B(WIDE8_M3_KTILE_ASM_B);
    else WIDE_STMT_B(WIDE_M3_KTILE_ASM_B);
  } else if (NF == 8)
    asm volatile(WIDE8_KTILE_ASM_B
                 : "=&v"(w1), "+v"(w0)
                 : "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(wa1)
                 : "memory", WIDE_AGPR_CLOBBERS);
  else
    asm volatile(WIDE_KTILE_ASM_B
                 : "=&v"(w1), "+v"(w0)
                 : "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(wa1)
                 : "memory", WIDE_AGPR_CLOBBERS);
#undef WIDE_STMT_B
}

// accumulator fragment (weight fragment i, token fragment j) out of the AGPRs; BASE = (4i + j) * 4
template <int BASE>
__device__ __forceinline__ float4_t wide_read_acc() {
  float a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\t"
               "v_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
               : "i"(BASE));
  return (float4_t){a, b, c, d};
}

template <int V> struct WideIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void wide_static_for(F&& f, WideIC<Is>...) { (f(WideIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void wide_for10(F&& f) {
  wide_static_for(f, WideIC<0>{}, WideIC<1>{}, WideIC<2>{}, WideIC<3>{}, WideIC<4>{}, WideIC<5>{}, WideIC<6>{}, WideIC<7>{}, WideIC<8>{}, WideIC<9>{});
}
template <class F> __device__ __forceinline__ void wide_for8(F&& f) {
  wide_static_for(f, WideIC<0>{}, WideIC<1>{}, WideIC<2>{}, WideIC<3>{}, WideIC<4>{}, WideIC<5>{}, WideIC<6>{}, WideIC<7>{});
}
template <int NF, class F> __device__ __forceinline__ void wide_for_nf(F&& f) {     // over a wave's weight fragments
  if constexpr (NF == 8) wide_for8(f);
  else wide_for10(f);
}
template <class F> __device__ __forceinline__ void wide_for5(F&& f) {
  wide_static_for(f, WideIC<0>{}, WideIC<1>{}, WideIC<2>{}, WideIC<3>{}, WideIC<4>{});
}

// SPLIT = false is the production instantiation: ksplit folds to 1 and the slice bookkeeping disappears (with it in, the
// allocator spilled 19-34 instead of 2-20 registers around the K-loop and every launch of this kernel got ~2 % slower)
// NF = weight fragments per wave: 10 = the 256x320 tile (wave tile 64 x 160), 8 = the 256x256 tile (64 x 128) for channel
// counts that are multiples of 256 and not of 320 (the VAE decoder's 256 / 512).  The LDS map is the 256x320 one in both
// forms (the weight part of a stage is filled to 32 of its 40 KiB), so is everything outside the constants below.
// MF = token fragments per wave: 4 = 256-row tiles (wave tile 64 rows), 3 = 192-row tiles (48 rows) for row counts whose
// 256-row tiles leave CUs idle (a sharded rank's levels): the same LDS map with the token part of a stage filled to 24 of
// its 32 KiB.
template <int MODE, bool SPLIT, bool LDSOUT, int NF, int MF>
__device__ __forceinline__ void wide_body(const lkgd_gemm_desc& p, int tiles_m, int tiles_n, int ksplit_arg, float* ws) {
  static_assert(NF == 10 || (NF == 8 && !SPLIT), "wave tiles: 64 x 160 or 64 x 128 (the latter unsliced)");
  static_assert(MF == 4 || (MF == 3 && !SPLIT), "wave tiles: 64 or 48 rows (the latter unsliced)");
  constexpr int WM = 64 * MF;       // tile rows
  constexpr int WRW = 16 * MF;      // a wave's rows
  constexpr int WN = 32 * NF;       // tile columns
  constexpr int WCH = 16 * NF;      // a wave's channels
  const int ksplit = SPLIT ? ksplit_arg : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- XCD-cooperative persistent schedule (see gemm_stream.hip)
  // split-K (few-row problems, ksplit > 1): the stream runs over VIRTUAL tiles vt = tile * ksplit + slice; a slice owns
  // K-tiles [slice * nk, (slice + 1) * nk) (equal slices: the launcher picks a divisor) and leaves its fp32 partial tile in
  // ws[slice][M][N] for lkgd_gemm_splitk_reduce.  Slices of a tile are adjacent in the stream: same XCD, shared panels.
  const int ntiles = tiles_m * tiles_n * ksplit;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, c = blockIdx.x >> 3;
  const int nc = (G - xcd + 7) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xb = xcd * q8 + (xcd < r8 ? xcd : r8);
  const int xe = xb + q8 + (xcd < r8 ? 1 : 0);
  const int my_tiles = (xe - xb - c + nc - 1) / nc;
  const int tile_begin = xb + c;
  const int nk = p.K / BK / ksplit;
  const int total = (my_tiles > 0 ? my_tiles : 0) * nk;
  if (total <= 0) return;
  // static priority for the later-dispatched half (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): waves 4-7 lose every
  // arbitration against their SIMD partner at equal priority; one s_setprio for the whole kernel, no per-phase flips.  Isolated
  // GEMM sum of a forward 71.50 -> 71.22 / 71.36 -> 71.08 ms, bench pairs 5.866 -> 5.884 / 5.880 -> 5.881 frames/s (round 5)
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
#ifdef WIDE_X_STAGGER
  for (int i = 0; i < (c & 3) * WIDE_X_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);   // 127 * 64 clocks each
#endif

  // ---- staging state: describes the K-tile whose loads are issued next
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  LeanGather<4> ag;           // 2-register row descriptors, division-free segment updates (gemm_common.h)
  float rcp0, rcp1;
  lean_rcps<MODE>(p, rcp0, rcp1);
  int st_tile = tile_begin - nc, st_kt = nk - 1, st_s = -1, st_par = 0, ep_par = 0;
  int st_k0 = 0;                // first K-tile of the staged virtual tile's slice
  ag.seg_k0 = 0; ag.seg_end = 0; ag.zmask = 0; ag.ky = 0; ag.kx = 0; ag.cc = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ag.rd[i].base = -1; ag.rd[i].yx = 0; ag.aptr[i] = (const half_t*)p.zeros; }
  WideIn in;
#pragma unroll
  for (int i = 0; i < 5; ++i) in.oB[i] = 0;
  // Row-indexed bias (time embedding / frame position tables): when the row map is piecewise constant over >= 256 rows
  // (idx = (m / d1) * m1 + c0 mod md with d1 >= 256 - every use but the cross-attention table), a tile's rows select at
  // most two table rows: those of its first and last token.  Both strips ride the LDS-DMA stream like the bias.
  const bool rb_lds = p.rowbias && p.rb_d2 == 1 && p.rb_d1 >= WBM;
  const half_t* wbase = (const half_t*)p.w;
  asm volatile("" : "+s"(wbase));            // an opaque SGPR pair: kept (or spilled to a lane), never re-loaded from kernarg
  // move to the next K-tile of the stream (stays on the last one at the end: the loads the uniform K-tile body issues there
  // re-read valid memory into the stage nobody reads any more) and return its DMA sources
  auto next_in = [&](WideIn& in) {
    if (st_s + 1 < total) {
      ++st_s;
      if (++st_kt == nk) {
        st_kt = 0;
        st_tile += nc;
        int tm, tn;
        supertile<4>(st_tile / ksplit, tiles_m, tiles_n, tm, tn);
        st_k0 = SPLIT ? (st_tile % ksplit) * nk : 0;
#pragma unroll
        for (int i = 0; i < MF; ++i) ag.rd[i] = lean_row<MODE>(p, WIDE_A_ROW(tm * WM) + srow + 64 * i, rcp0, rcp1);   // (rows past MF stay on the zero page)
#pragma unroll
        for (int i = 0; i < NF / 2; ++i) {
          int n = tn * WN + srow + 64 * i;
          n = n < p.N ? n : p.N - 1;     // clamped: channels past N are computed on a copy of the last row, never stored
          in.oB[i] = ((unsigned)n * (unsigned)p.K + schunk * 8) * 2u;
        }
        ag.seg_end = 0;
        // the tile's bias strip rides the LDS-DMA stream too (4 bytes per lane, waves 0-4): the epilogue then reads it with
        // ds_read_b128 instead of waiting one global-load latency per token fragment
        st_par ^= 1;
        if (p.bias && w < NF / 2) {
          int n = tn * WN + w * 64 + lane;
          n = n < p.N ? n : p.N - 1;
          __builtin_amdgcn_global_load_lds(GLB_PTR(p.bias + n), LDS_PTR(smem + WBIAS_OFF + st_par * (WBN * 4) + w * 256), 4, 0, 0);
        }
        if (rb_lds && w >= 5) {
          // waves 5-7 fetch dwords (w-5)*64 + lane of the two 160-dword strips (lanes past 160 re-read the last dword
          // into the strip's padding)
          const unsigned mf = (unsigned)(tm * WM);
          unsigned ml = mf + WM - 1;
          ml = ml < (unsigned)p.M ? ml : (unsigned)p.M - 1;
          const unsigned i0 = ((mf / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          const unsigned i1 = ((ml / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          int dw = (w - 5) * 64 + lane;
          int col = tn * WN + 2 * (dw < WN / 2 ? dw : WN / 2 - 1);
          col = col < p.N - 2 ? col : p.N - 2;
          char* dst = smem + WRB_OFF + st_par * (2 * WRB_STRIP) + (w - 5) * 256;
          const half_t* rb = (const half_t*)p.rowbias;
          __builtin_amdgcn_global_load_lds(GLB_PTR(rb + (long long)i0 * p.ldrb + col), LDS_PTR(dst), 4, 0, 0);
          __builtin_amdgcn_global_load_lds(GLB_PTR(rb + (long long)i1 * p.ldrb + col), LDS_PTR(dst + WRB_STRIP), 4, 0, 0);
        }
      }
      if ((st_k0 + st_kt) * BK >= ag.seg_end) {
        lean_segment<MODE, 4>(p, ag, (st_k0 + st_kt) * BK, schunk);   // aptr = the rows' sources at the segment's first K-tile
        const int into = SPLIT ? (st_k0 + st_kt) * BK - ag.seg_k0 : 0;   // a K slice may start inside a segment
        if (SPLIT && into) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ag.aptr[i] += ((ag.zmask >> i) & 1u) ? 0 : into;
        }
      } else {
        // within a segment a row's source advances by one K-tile (128 bytes); zero-page rows stay
#pragma unroll
        for (int i = 0; i < 4; ++i) ag.aptr[i] += ((ag.zmask >> i) & 1u) ? 0 : BK;
      }
    }
    in.wk = wbase + (st_k0 + st_kt) * BK;
  };

  // fragment rows: tokens wr*64 + mi*16 + l15, weights wc*160 + ni*16 + l15.  The swizzle key (row>>1)&7 is the same
  // for every fragment of a lane (16*mi, 16*ni, 64*wr, 160*wc are all 0 mod 16), so every fragment address is
  // base + compile-time offset: two chunk offsets (one per k-step) per lane
  const int skey = (l15 >> 1) & 7;
  int xa0, xa1, wa0, wa1;       // LDS addresses of fragment 0 (tokens / weights, K-step 0 / 1) in the CURRENT stage
  {
    const int x_base = (wr * WRW + l15) * 128;
    const int w_base = WBM * BK * 2 + (wc * WCH + l15) * 128;
    const int ch0 = ((0 + lq) ^ skey) << 4, ch1 = ((4 + lq) ^ skey) << 4;
    xa0 = x_base + ch0; xa1 = x_base + ch1; wa0 = w_base + ch0; wa1 = w_base + ch1;
  }

  // ---- K-tile 0 -> stage 0
  {
    next_in(in);
    char* sx = smem + w * 1024;
#pragma unroll
    for (int i = 0; i < MF; ++i) glds16(ag.aptr[i], sx + 8192 * i);
#pragma unroll
    for (int i = 0; i < NF / 2; ++i) glds16((const half_t*)((const char*)in.wk + in.oB[i]), sx + WBM * BK * 2 + 8192 * i);
  }
  next_in(in);                  // K-tile 1

  int cur = 0, kt = 0, tile = tile_begin;
  bool skip_wait = false;
#ifdef WIDE_X_STAMPS       /* diagnostic build (tools/micro/wide_stamps.py): s_memtime sums per wave, written over the first output rows */
  long long st_sync = 0, st_body = 0, st_epi = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
#endif
#ifdef WIDE_X_LATEWAIT     /* timing knob (results wrong): the second body after an epilogue leaves its stores in flight */
  int late = 0;
#endif
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    // K-tile s must have landed (it is the only LDS-DMA batch in flight at this point); the K-tile body starts with the
    // workgroup barrier, issues K-tile s+1's loads into the other stage and computes K-tile s
#ifdef WIDE_X_LATEWAIT
    if (late == 1) { asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); skip_wait = true; }
    late = skip_wait && late == 0 ? 1 : 0;
#endif
#ifdef WIDE_X_STAMPS
    st_a = __builtin_amdgcn_s_memtime();
#endif
    if (!skip_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    skip_wait = false;
#ifdef WIDE_X_STAMPS
    __builtin_amdgcn_s_barrier();
    st_b = __builtin_amdgcn_s_memtime(); st_sync += st_b - st_a;
#endif
    const int m_a = (cur ^ 1) * WSTAGE_BYTES + w * 1024;
    half8_t x1[4], w0;
    if (kt == 0) wide_ktile_a<true, NF, MF>(x1, w0, xa0, xa1, wa0, wa1, ag.aptr, in, m_a);
    else wide_ktile_a<false, NF, MF>(x1, w0, xa0, xa1, wa0, wa1, ag.aptr, in, m_a);
    const int wa1_now = wa1;
    next_in(in);                // K-tile s+2's sources, for the next body
    wide_ktile_b<NF, MF>(x1, w0, wa1_now);
    {
      const int d = cur ? -WSTAGE_BYTES : WSTAGE_BYTES;      // the other stage becomes the current one
      xa0 += d; xa1 += d; wa0 += d; wa1 += d;
    }
    cur ^= 1;
#ifdef WIDE_X_STAMPS
    st_a = __builtin_amdgcn_s_memtime(); st_body += st_a - st_b;
#endif

    if (++kt == nk) {
      // ---------------------------------------------------------------- epilogue of `tile`, straight from registers.
      // Take step s+1's wait first (only K-tile s+1 is outstanding), so epilogue traffic never sits before it.
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // + the last MFMAs have left the pipe
      skip_wait = true;
      kt = 0;
      int lane_e = lane;             // opaque copy: everything per-lane below is recomputed here, once per tile, instead of
      asm volatile("" : "+v"(lane_e));   // being hoisted out of the persistent loop into registers the K-tile body needs
      const int l15 = lane_e & 15, lq = lane_e >> 4;      // (shadow the kernel-scope ones)
      int tm, tn;
      supertile<4>(tile / ksplit, tiles_m, tiles_n, tm, tn);
      const int slice = tile % ksplit;
      tile += nc;
      const int m0 = tm * WM + wr * WRW + l15;
      const int n0 = tn * WN + wc * WCH + 4 * lq;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      // one token fragment (16 tokens x 160 channels) at a time
      ep_par ^= 1;
      const float* bl = (const float*)(smem + WBIAS_OFF + ep_par * (WBN * 4)) + wc * WCH + 4 * lq;   // bias[n0 + ...]
      const half_t* rbl = (const half_t*)(smem + WRB_OFF + ep_par * (2 * WRB_STRIP)) + wc * WCH + 4 * lq;
      // rows below rb_bound use the first strip, the others the second (the map changes at most once inside the tile)
      const unsigned rb_bound = rb_lds ? ((unsigned)(tm * WM) / (unsigned)p.rb_d1 + 1u) * (unsigned)p.rb_d1 : 0u;
      // Output rows through LDS (LDSOUT: the plain linears without GEGLU).  In the accumulator layout a lane holds 4 channels
      // of ONE token row, so a direct store instruction writes 16 rows x 32 bytes; a CU sustains ~16 GB/s of those against
      // 24-60 GB/s for 16-byte pieces of contiguous rows (tools/micro/store_bw.hip), and the short-K linears are store-bound
      // (stores compiled out: -34 % on the K = 320 projections of the 72x128 level).  The K-tile stage consumed last is free
      // until the next body's loads (one barrier makes sure every wave is done reading it): each wave parks a token fragment
      // (16 rows x 160 channels fp16, 336-byte pitch) there and writes it out as 16-byte pieces of 320-byte row segments.
      // Measured (profiles/r02_gemm_ldsout_ab.txt): +1-6 % on the plain linears, -2 % on the deep-K convolutions (the extra
      // barrier and LDS pass cost more than their rare epilogue returns), -4-10 % with GEGLU (160-byte segments): used for
      // the first only.
      constexpr bool lds_out = LDSOUT && !SPLIT;        // the launcher checks N % 320 == 0, ldc % 8 == 0, 16-byte aligned out
      char* scr = smem + (cur ^ 1) * WSTAGE_BYTES + w * (16 * WIDE_OUT_PITCH);
      if (lds_out) __builtin_amdgcn_s_barrier();
      // Column statistics (lkgd_gemm_desc.colstats: the GroupNorm sums of the tensor this GEMM writes).  A token fragment
      // sits in LDS as 16 row segments of the wave's 160 channels before it is stored: lanes 0-39 add up four channels each
      // over the rows (the ROUNDED fp16 values the consumer will read), carry the sums over the tile's four fragments, and
      // the four waves of a channel half are combined in a fixed order at the end of the epilogue - no atomics.
      const bool cs_on = lds_out && p.colstats != nullptr;
      float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};      // this lane's two channel PAIRS (GroupNorm groups are even-sized)
      // accumulator fragments are read where they are used (one or two live at a time, not all ten of a token fragment)
      auto epi = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const long long m = m0 + j * 16;
        const bool live = m < p.M;
        if (!lds_out && !live) return;
        if (SPLIT) {                 // fp32 partial tile of this K slice; bias / residuals / rounding happen in the reduce pass
          float* dst = ws + ((long long)slice * p.M + m) * p.N;
          wide_for_nf<NF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float4_t ev = wide_read_acc<(4 * i + j) * 4>();
            if (n0 + i * 16 < p.N) *(float4_t*)(dst + n0 + i * 16) = ev;
          });
          return;
        }
        const bool gg = NF == 10 && MODE == LKGD_A_PLAIN && !lds_out && p.geglu;      // (the launcher never pairs GEGLU with LDSOUT, nor with NF = 8)
        if (!gg) {
          unsigned idx = 0;          // 32-bit row-map arithmetic: M < 2^24 is a launch condition of this kernel
          if (rbp && !rb_lds && live) idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) +
                                             (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          const int rb_sel = (unsigned)m < rb_bound ? 0 : WRB_STRIP / 2;
          wide_for_nf<NF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int n = n0 + i * 16;
            float4_t v = wide_read_acc<(4 * i + j) * 4>();
            if (n < p.N) {
              if (p.bias) v += *(const float4_t*)(bl + i * 16);
              if (rbp && live) {
                const half4_t rb = rb_lds ? *(const half4_t*)(rbl + rb_sel + i * 16)
                                          : *(const half4_t*)(rbp + (long long)idx * p.ldrb + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += (float)rb[x];
              }
              v *= p.s_acc;
              if (r1p && live) {
                half4_t r = *(const half4_t*)(r1p + m * p.ldr1 + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += p.r1 * (float)r[x];
              }
              if (r2p && live) {
                half4_t r = *(const half4_t*)(r2p + m * p.ldr2 + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += p.r2 * (float)r[x];
              }
              half4_t o;
#pragma unroll
              for (int x = 0; x < 4; ++x) o[x] = (half_t)v[x];
              if (lds_out) *(half4_t*)(scr + l15 * WIDE_OUT_PITCH + (i * 16 + 4 * lq) * 2) = o;
              else { WIDE_STORE_GUARD WIDE_ST((half4_t*)(outp + WIDE_OUT_ROW(m) * p.ldc + n), o); }
            }
          });
        } else {
          // wave channels [0,80) = hidden, [80,160) = gate of output columns tn*160 + wc*80 + [0,80)
          const int oc0 = tn * 160 + wc * 80 + 4 * lq;
          wide_for5([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            float4_t hv = wide_read_acc<(4 * i + j) * 4>(), gv = wide_read_acc<(4 * (i + 5) + j) * 4>();
            if (p.bias) {
              hv += *(const float4_t*)(bl + i * 16);
              gv += *(const float4_t*)(bl + 80 + i * 16);
            }
#ifdef WIDE_X_NOGELU     /* timing knob: the gate without its GELU */
            const float2_t lo = __builtin_shufflevector(hv, hv, 0, 1) * __builtin_shufflevector(gv, gv, 0, 1);
            const float2_t hi = __builtin_shufflevector(hv, hv, 2, 3) * __builtin_shufflevector(gv, gv, 2, 3);
#else
            const float2_t lo = __builtin_shufflevector(hv, hv, 0, 1) * gelu_erf2(__builtin_shufflevector(gv, gv, 0, 1));
            const float2_t hi = __builtin_shufflevector(hv, hv, 2, 3) * gelu_erf2(__builtin_shufflevector(gv, gv, 2, 3));
#endif
            const half4_t o = {(half_t)lo.x, (half_t)lo.y, (half_t)hi.x, (half_t)hi.y};
            if (lds_out) *(half4_t*)(scr + l15 * WIDE_OUT_PITCH + (i * 16 + 4 * lq) * 2) = o;
            else { WIDE_STORE_GUARD WIDE_ST((half4_t*)(outp + WIDE_OUT_ROW(m) * p.ldc + oc0 + i * 16), o); }
          });
        }
        if (lds_out) {
          // the fragment's rows as 16-byte pieces: piece c of the wave's 16 x (20 | 10) grid -> row c / per, column piece c % per
          const int per = gg ? 10 : 2 * NF;                           // 16-byte pieces per row (80 | 160 | 128 channels)
          const long long mrow0 = (long long)tm * WM + wr * WRW + j * 16;
          if (cs_on && lane_e < 4 * NF) {
            // branch-free over the 16 rows (rows past M are weighted 0): the reads issue back to back; per channel PAIR
            // one v_dot2_f32_f16 for the sum and one for the sum of squares (fp16 products are exact in fp32)
            const long long left = (long long)p.M - mrow0;
            const int nrows = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
            #pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {      // eight rows in flight at a time (16 registers)
              half4_t u[8];
#pragma unroll
              for (int r = 0; r < 8; ++r) u[r] = *(const half4_t*)(scr + (r0 + r) * WIDE_OUT_PITCH + lane_e * 8);
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
          const int ncol0 = gg ? tn * 160 + wc * 80 : tn * WN + wc * WCH;
#pragma unroll
          for (int kk = 0; kk < 5; ++kk) {
            const int c = lane_e + 64 * kk;
            if (c < 16 * per) {
              const int row = NF == 8 ? c >> 4 : (gg ? (c * 205) >> 11 : (c * 205) >> 12);   // c / 16 | c / 10, c / 20 for c < 320
              const int cc = c - row * per;
              const half8_t v = *(const half8_t*)(scr + row * WIDE_OUT_PITCH + cc * 16);
              const long long mr = mrow0 + row;
              if (mr < p.M) { WIDE_STORE_GUARD WIDE_ST((half8_t*)(outp + WIDE_OUT_ROW(mr) * p.ldc + ncol0 + cc * 8), v); }
            }
          }
        }
      };
#define WIDE_EPI(J)                                                                                                \
  {                                                                                                                \
    epi(WideIC<J>{});                                                                                              \
    __builtin_amdgcn_sched_barrier(0);  /* keep one token fragment's loads/stores from piling onto the next */   \
  }
      WIDE_EPI(0) WIDE_EPI(1) WIDE_EPI(2)
      if constexpr (MF == 4) WIDE_EPI(3)
#undef WIDE_EPI
      if (cs_on) {
        // [row group wr][160 channel pairs][sum, sum of squares] behind the eight waves' row patches in the free stage; the
        // next K-tile body starts with a barrier before anything is loaded into this stage
        float* part = (float*)(smem + (cur ^ 1) * WSTAGE_BYTES + 8 * 16 * WIDE_OUT_PITCH);
        if (lane_e < 4 * NF)
          *(float4_t*)(part + (wr * (WN / 2) + wc * (WCH / 2) + 2 * lane_e) * 2) = (float4_t){csum[0], csq[0], csum[1], csq[1]};
        __builtin_amdgcn_s_barrier();
        if (t < WN / 2) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) { a += part[(g * (WN / 2) + t) * 2]; b += part[(g * (WN / 2) + t) * 2 + 1]; }
          typedef float float2v __attribute__((ext_vector_type(2)));
          *(float2v*)(p.colstats + ((long long)tm * (p.N / 2) + tn * (WN / 2) + t) * 2) = (float2v){a, b};
        }
      }
#ifdef WIDE_X_STAMPS
      st_epi += __builtin_amdgcn_s_memtime() - st_a;
#endif
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's trailing loads must land before the LDS is released
#ifdef WIDE_X_STAMPS
  if (lane == 0) {
    long long* dbg = (long long*)p.out + ((long long)blockIdx.x * 8 + w) * 4;
    dbg[0] = st_sync; dbg[1] = st_body; dbg[2] = st_epi; dbg[3] = __builtin_amdgcn_s_memtime() - st_t0;
  }
#endif
}

template <int MODE, bool LDSOUT, int NF = 10, int MF = 4>
__global__ __launch_bounds__(WNT, 2) __attribute__((amdgpu_num_vgpr(96))) void lkgd_gemm_wide_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  wide_body<MODE, false, LDSOUT, NF, MF>(p, tiles_m, tiles_n, 1, nullptr);
}
template <int MODE>
__global__ __launch_bounds__(WNT, 2) __attribute__((amdgpu_num_vgpr(96))) void lkgd_gemm_wide_split_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n, int ksplit, float* ws) {
  wide_body<MODE, true, false, 10, 4>(p, tiles_m, tiles_n, ksplit, ws);
}

// ksplit > 1: K is cut into ksplit equal slices (ksplit divides K / 64); the caller runs lkgd_gemm_splitk_reduce afterwards
static thread_local int wide_lds_out_override = -1;     // A/B knob: 0 = direct 8-byte stores everywhere, 1 / -1 = rows through LDS where used
extern "C" void lkgd_debug_set_wide_lds_out(int on) { wide_lds_out_override = on; }

// the 256x256 form serves channel counts that are whole 256-column tiles and not whole 320-column ones (256, 512, 768 ...)
// and would leave more than a tenth of the 320-wide tile columns idle (3072 = 9.6 x 320 stays on 320)
static thread_local int wide_tile_n_forced = 0;      // A/B knob (tools/micro/wide_tile_n.py): 256 / 320 where that width divides N, 0 = the rule
extern "C" void lkgd_debug_set_wide_tile_n(int wn) { wide_tile_n_forced = (wn == 256 || wn == 320) ? wn : 0; }
extern "C" int lkgd_debug_wide_tile_n_forced() { return wide_tile_n_forced; }
static thread_local int wide_tile_m_forced = 0;      // A/B knob: 192 / 256 tile rows (0 = the rule of gemm.hip::gemm_wide_form)
extern "C" void lkgd_debug_set_wide_tile_m(int wm) { wide_tile_m_forced = (wm == 192 || wm == 256) ? wm : 0; }
extern "C" int lkgd_debug_wide_tile_m_forced() { return wide_tile_m_forced; }
extern "C" int lkgd_gemm_wide_tile_n(int N) {
  if (wide_tile_n_forced && N % wide_tile_n_forced == 0) return wide_tile_n_forced;
  return (N % WBN != 0 && N % 256 == 0 && (long long)((N + WBN - 1) / WBN) * WBN * 10 > (long long)N * 11) ? 256 : WBN;
}

template <int MODE>
static void wide_go(const lkgd_gemm_desc* d, hipStream_t stream, int grid, int tiles_m, int tiles_n, bool lds_out, int wn, int wm) {
#define WIDE_GO(...) \
  hipLaunchKernelGGL((lkgd_gemm_wide_kernel<MODE, __VA_ARGS__>), dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n)
  if (wm == 192) {
    if (wn == 256) { if (lds_out) WIDE_GO(true, 8, 3); else WIDE_GO(false, 8, 3); }
    else { if (lds_out) WIDE_GO(true, 10, 3); else WIDE_GO(false, 10, 3); }
  } else {
    if (wn == 256) { if (lds_out) WIDE_GO(true, 8, 4); else WIDE_GO(false, 8, 4); }
    else { if (lds_out) WIDE_GO(true, 10, 4); else WIDE_GO(false, 10, 4); }
  }
#undef WIDE_GO
}
template <int MODE>
static bool wide_set_lds(const void* split_fn) {
  const void* fns[] = {(const void*)lkgd_gemm_wide_kernel<MODE, false, 10, 4>, (const void*)lkgd_gemm_wide_kernel<MODE, true, 10, 4>,
                       (const void*)lkgd_gemm_wide_kernel<MODE, false, 8, 4>,  (const void*)lkgd_gemm_wide_kernel<MODE, true, 8, 4>,
                       (const void*)lkgd_gemm_wide_kernel<MODE, false, 10, 3>, (const void*)lkgd_gemm_wide_kernel<MODE, true, 10, 3>,
                       (const void*)lkgd_gemm_wide_kernel<MODE, false, 8, 3>,  (const void*)lkgd_gemm_wide_kernel<MODE, true, 8, 3>, split_fn};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS) != hipSuccess) return false;
  return true;
}

// wn = tile columns (320 | 256), wm = tile rows (256 | 192): the caller's choice (gemm.hip: gemm_wide_form)
extern "C" int lkgd_gemm_wide_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus, int ksplit, int wn, int wm) {
  LKGD_DEVICE_ONCE_BEGIN
    if (!wide_set_lds<LKGD_A_PLAIN>((const void*)lkgd_gemm_wide_split_kernel<LKGD_A_PLAIN>) ||
        !wide_set_lds<LKGD_A_CONV3X3>((const void*)lkgd_gemm_wide_split_kernel<LKGD_A_CONV3X3>) ||
        !wide_set_lds<LKGD_A_TCONV3>((const void*)lkgd_gemm_wide_split_kernel<LKGD_A_TCONV3>))
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  if ((wn != 256 && wn != WBN) || (wm != 192 && wm != WBM)) return LKGD_E_SHAPE;
  int tiles_m = (d->M + wm - 1) / wm, tiles_n = (d->N + wn - 1) / wn;
  if (ksplit < 1 || (d->K / BK) % ksplit || (ksplit > 1 && (!d->workspace || d->geglu))) return LKGD_E_SHAPE;
  if ((wn == 256 || wm == 192) && ksplit > 1) return LKGD_E_SHAPE;      // K slices: 256x320 tiles only
  if (wn == 256 && d->geglu) return LKGD_E_SHAPE;                       // the 80-wide GEGLU interleave needs 320-column tiles
  long long ntiles = (long long)tiles_m * tiles_n * ksplit;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  if (d->M >= (1 << 24)) return LKGD_E_SHAPE;            // float-reciprocal row decomposition (gemm_common.h)
  float* ws = (float*)d->workspace;
  // plain linears without GEGLU, whole tile columns, 16-byte aligned output rows: the rows leave through LDS (see the epilogue)
  // (the convolutions only when column statistics are asked for: they are summed from the rows in LDS)
  const bool lds_ok = ksplit == 1 && !d->geglu && d->N % wn == 0 && d->ldc % 8 == 0 && aligned16(d->out);
  if (d->colstats && !lds_ok) return LKGD_E_SHAPE;
  const bool lds_out = lds_ok && (d->colstats ? true : (wide_lds_out_override != 0 && d->mode == LKGD_A_PLAIN));
#define WIDE_LAUNCH(MODE_)                                                                                              \
  {                                                                                                                     \
    if (ksplit > 1)                                                                                                     \
      hipLaunchKernelGGL(lkgd_gemm_wide_split_kernel<MODE_>, dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n, ksplit, ws); \
    else                                                                                                                \
      wide_go<MODE_>(d, stream, grid, tiles_m, tiles_n, lds_out, wn, wm);                                               \
  }
  if (d->mode == LKGD_A_PLAIN) WIDE_LAUNCH(LKGD_A_PLAIN)
  else if (d->mode == LKGD_A_CONV3X3) WIDE_LAUNCH(LKGD_A_CONV3X3)
  else if (d->mode == LKGD_A_TCONV3) WIDE_LAUNCH(LKGD_A_TCONV3)
#undef WIDE_LAUNCH
  else
    return LKGD_E_MODE;
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
