// 256x320-tile MFMA GEMM / implicit convolution on FOUR waves of 128 tokens x 160 channels (one wave per SIMD, 512
// registers each) - the low-LDS-traffic sibling of gemm_wide.hip (eight waves of 64 x 160): per 80 MFMAs a wave reads
// 8 + 10 operand fragments instead of 2 x (4 + 10), and the per-K-tile barrier joins four waves instead of eight.
// Same tile, LDS image (two 72 KiB stages + bias / row-bias strips), lean gather, persistent XCD-cooperative schedule and
// epilogues as gemm_wide.hip; the differences are the thread -> row maps (32 staging rows per LDS-DMA piece group, eight
// A rows + ten weight rows per thread), the generated K-tile body (tools/gen_wide4_asm.py: accumulators in a[0:255] +
// v[192:255], fragments in v[116:191], the compiler confined to v[0:115]) and the accumulator read-out.
#include "gemm_common.h"

#define WBM 256
#define WBN 320
#define WNT 256
#define WSTAGE_BYTES ((WBM + WBN) * BK * 2)   // 72 KiB
#define WBIAS_OFF (2 * WSTAGE_BYTES)          // two 320-float bias strips (tile parity) behind the stages
#define WRB_OFF (WBIAS_OFF + 2 * WBN * 4)      // row-bias strips: [tile parity][first | last row's table row][384 halfs]
#define WRB_STRIP 768
#define WLDS (WRB_OFF + 4 * WRB_STRIP)        // 149.5 KiB

// ---- K-tile body (generated asm, tools/gen_wide4_asm.py)
#define WIDE4_WAIT_TOP "s_waitcnt lgkmcnt(0)\n\t"   /* no scalar load of the surrounding code in the counted LDS waits */
// (timing-knob builds, tools/micro/wide_knobs.sh, drop one ingredient; never defined in the product build)
#ifdef WIDE_X_NOSTAGE
#define WIDE4_LD(X) ""
#else
#define WIDE4_LD(X) X
#endif
#ifdef WIDE_X_NOREAD
#define WIDE4_RD(X) ""
#else
#define WIDE4_RD(X) X
#endif
#ifdef WIDE_X_NOMFMA
#define WIDE4_MM(X) ""
#else
#define WIDE4_MM(X) X
#endif
#ifdef WIDE_X_NOBAR
#define WIDE4_BAR(X) ""
#else
#define WIDE4_BAR(X) X
#endif
#define WIDE_STORE_GUARD
#define WIDE_OUT_ROW(m) (m)
#define WIDE_ST(ptr, v) (*(ptr) = (v))
#include "gemm_wide4_ktile.inc"

struct WideIn {                 // what one K-tile body needs about the NEXT K-tile (its nine DMA loads), besides the
  unsigned oB[10];              // A row sources in LeanGather::aptr: byte offsets of the ten weight row chunks from wk
  const half_t* wk;             // weights + K offset (wave-uniform)
};

// statement A: barrier, K-step 0 (80 MFMAs), the eight A-row LDS-DMA loads of the next K-tile (one per weight-fragment
// step), prefetch of K-step 1's fragments.  Fragments and accumulators live in registers the compiler does not allocate
// (WIDE4_CLOBBERS): inputs only.
template <bool FIRST>
__device__ __forceinline__ void wide4_ktile_a(int xa0, int xa1, int wa0, int wa1, const half_t* const (&pA)[8], int m_a) {
#define WIDE4_STMT(BODY)                                                                                               \
  asm volatile(BODY                                                                                                    \
               :                                                                                                       \
               : "v"(xa0), "v"(xa1), "v"(wa0), "v"(wa1), "v"(pA[0]), "v"(pA[1]), "v"(pA[2]), "v"(pA[3]), "v"(pA[4]),   \
                 "v"(pA[5]), "v"(pA[6]), "v"(pA[7]), "s"(m_a)                                                          \
               : "memory", "scc", WIDE4_CLOBBERS)
  if (FIRST) {
    WIDE4_STMT(WIDE4_KTILE_ASM_FIRST_A);
  } else {
    WIDE4_STMT(WIDE4_KTILE_ASM_NEXT_A);
  }
#undef WIDE4_STMT
}

// statement B: K-step 1 (80 MFMAs) and the ten weight-row loads.  Between A and B the C++ advances the A-row sources to the
// K-tile after next (next_a) while K-step 0's MFMAs drain; the weight sources advance after B (next_w: scalar but for a
// tile change).
// (WIDE4_TAIL: wa0n / xa0n = fragment-0 addresses of the OTHER stage: the statement ends with the K-tile boundary - vmcnt
// wait, barrier, the next K-tile's first reads - in front of its last eight MFMAs)
__device__ __forceinline__ void wide4_ktile_b(int wa1, const WideIn& in, int m_a, int wa0n, int xa0n) {
  asm volatile(WIDE4_KTILE_ASM_B
               :
               : "v"(wa1), "v"(in.oB[0]), "v"(in.oB[1]), "v"(in.oB[2]), "v"(in.oB[3]), "v"(in.oB[4]), "v"(in.oB[5]),
                 "v"(in.oB[6]), "v"(in.oB[7]), "v"(in.oB[8]), "v"(in.oB[9]), "s"(in.wk), "s"(m_a), "v"(wa0n), "v"(xa0n)
               : "memory", "scc", WIDE4_CLOBBERS);
}

// accumulator block (weight fragment I, token fragment J): AGPRs for I < 8, pinned VGPRs above
template <int I, int J>
__device__ __forceinline__ float4_t wide4_read_acc() {
  float a, b, c, d;
  if constexpr (I < 8) {
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\t"
                 "v_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
                 : "i"((8 * I + J) * 4));
  } else {
    asm volatile("v_mov_b32 %0, v[%4]\n\tv_mov_b32 %1, v[%4+1]\n\tv_mov_b32 %2, v[%4+2]\n\tv_mov_b32 %3, v[%4+3]"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
                 : "i"(192 + ((I - 8) * 8 + J) * 4));
  }
  return (float4_t){a, b, c, d};
}

template <int MODE>
__global__ __launch_bounds__(WNT, 1) __attribute__((amdgpu_num_vgpr(WIDE4_VC))) void lkgd_gemm_wide4_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- XCD-cooperative persistent schedule (see gemm_stream.hip)
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
  const int total = (my_tiles > 0 ? my_tiles : 0) * nk;
  if (total <= 0) return;

  // ---- staging state: describes the K-tile whose loads are issued next
  const int srow = t >> 3;
  const int schunk = (t & 7) ^ ((t >> 4) & 7);
  LeanGather<8> ag;           // 2-register row descriptors, division-free segment updates (gemm_common.h)
  float rcp0, rcp1;
  lean_rcps<MODE>(p, rcp0, rcp1);
  int st_tile = tile_begin - nc, st_kt = nk - 1, st_s = -1, st_par = 0, ep_par = 0;
  ag.seg_k0 = 0; ag.seg_end = 0; ag.zmask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ag.rd[i].base = -1; ag.rd[i].yx = 0; ag.aptr[i] = (const half_t*)p.zeros; }
  WideIn in;
#pragma unroll
  for (int i = 0; i < 10; ++i) in.oB[i] = 0;
  // Row-indexed bias (time embedding / frame position tables): when the row map is piecewise constant over >= 256 rows
  // (idx = (m / d1) * m1 + c0 mod md with d1 >= 256 - every use but the cross-attention table), a tile's rows select at
  // most two table rows: those of its first and last token.  Both strips ride the LDS-DMA stream like the bias.
  const bool rb_lds = p.rowbias && p.rb_d2 == 1 && p.rb_d1 >= WBM;
  const half_t* wbase = (const half_t*)p.w;
  asm volatile("" : "+s"(wbase));            // an opaque SGPR pair: kept (or spilled to a lane), never re-loaded from kernarg
  // move to the next K-tile of the stream (stays on the last one at the end: the loads the uniform K-tile body issues there
  // re-read valid memory into the stage nobody reads any more) and return its DMA sources
  int st_tn = 0;
  bool st_new = false;          // next_a moved to a new tile: next_w owes the tile's weight-row offsets
  auto next_a = [&]() {
    if (st_s + 1 < total) {
      ++st_s;
      if (++st_kt == nk) {
        st_kt = 0;
        st_tile += nc;
        int tm, tn;
        supertile<4>(st_tile, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 8; ++i) ag.rd[i] = lean_row<MODE>(p, tm * WBM + srow + 32 * i, rcp0, rcp1);
        st_tn = tn;
        st_new = true;
        ag.seg_end = 0;
        // the tile's bias strip rides the LDS-DMA stream too (4 bytes per lane, waves 0-4): the epilogue then reads it with
        // ds_read_b128 instead of waiting one global-load latency per token fragment
        st_par ^= 1;
        // eleven 256-byte pieces (5 of the bias strip, 3 + 3 of the two row-bias strips): wave w takes pieces w, w+4, w+8
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const int piece = w + 4 * pc;
          if (piece < 5) {
            if (p.bias) {
              int n = tn * WBN + piece * 64 + lane;
              n = n < p.N ? n : p.N - 1;
              __builtin_amdgcn_global_load_lds(GLB_PTR(p.bias + n), LDS_PTR(smem + WBIAS_OFF + st_par * (WBN * 4) + piece * 256), 4, 0, 0);
            }
          } else if (piece < 11 && rb_lds) {
            const int strip = (piece - 5) / 3, part = (piece - 5) % 3;
            const unsigned mf = (unsigned)(tm * WBM);
            unsigned ml = mf + WBM - 1;
            ml = ml < (unsigned)p.M ? ml : (unsigned)p.M - 1;
            const unsigned mm = strip ? ml : mf;
            const unsigned ix = ((mm / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (unsigned)p.rb_c0) % (unsigned)p.rb_md;
            const int dw = part * 64 + lane;
            int col = tn * WBN + 2 * (dw < 160 ? dw : 159);
            col = col < p.N - 2 ? col : p.N - 2;
            char* dst = smem + WRB_OFF + st_par * (2 * WRB_STRIP) + strip * WRB_STRIP + part * 256;
            __builtin_amdgcn_global_load_lds(GLB_PTR((const half_t*)p.rowbias + (long long)ix * p.ldrb + col), LDS_PTR(dst), 4, 0, 0);
          }
        }
      }
      if (st_kt * BK >= ag.seg_end) {
        lean_segment<MODE, 8>(p, ag, st_kt * BK, schunk);       // aptr = the rows' sources at the segment's first K-tile
      } else {
        // within a segment a row's source advances by one K-tile (128 bytes); zero-page rows stay
#pragma unroll
        for (int i = 0; i < 8; ++i) ag.aptr[i] += ((ag.zmask >> i) & 1u) ? 0 : BK;
      }
    }
  };
  auto next_w = [&](WideIn& in) {
    if (st_new) {
      st_new = false;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        int n = st_tn * WBN + srow + 32 * i;
        n = n < p.N ? n : p.N - 1;     // clamped: channels past N are computed on a copy of the last row, never stored
        in.oB[i] = ((unsigned)n * (unsigned)p.K + schunk * 8) * 2u;
      }
    }
    in.wk = wbase + st_kt * BK;
  };

  // fragment rows: tokens wr*64 + mi*16 + l15, weights wc*160 + ni*16 + l15.  The swizzle key (row>>1)&7 is the same
  // for every fragment of a lane (16*mi, 16*ni, 64*wr, 160*wc are all 0 mod 16), so every fragment address is
  // base + compile-time offset: two chunk offsets (one per k-step) per lane
  const int skey = (l15 >> 1) & 7;
  int xa0, xa1, wa0, wa1;       // LDS addresses of fragment 0 (tokens / weights, K-step 0 / 1) in the CURRENT stage
  {
    const int x_base = (wr * 128 + l15) * 128;
    const int w_base = WBM * BK * 2 + (wc * 160 + l15) * 128;
    const int ch0 = ((0 + lq) ^ skey) << 4, ch1 = ((4 + lq) ^ skey) << 4;
    xa0 = x_base + ch0; xa1 = x_base + ch1; wa0 = w_base + ch0; wa1 = w_base + ch1;
  }

  // ---- K-tile 0 -> stage 0
  {
    next_a();
    next_w(in);
    char* sx = smem + w * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(ag.aptr[i], sx + 4096 * i);
#pragma unroll
    for (int i = 0; i < 10; ++i) glds16((const half_t*)((const char*)in.wk + in.oB[i]), sx + WBM * BK * 2 + 4096 * i);
  }
  next_a();                     // K-tile 1
  next_w(in);

#if WIDE4_TAIL
  asm volatile(WIDE4_KTILE_ASM_PRO : : "v"(xa0), "v"(wa0) : "memory", WIDE4_CLOBBERS);   // K-tile 0 landed, first reads out
#endif
  int cur = 0, kt = 0, tile = tile_begin;
  bool skip_wait = false;
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    // K-tile s must have landed (it is the only LDS-DMA batch in flight at this point); the K-tile body starts with the
    // workgroup barrier, issues K-tile s+1's loads into the other stage and computes K-tile s
#if !WIDE4_TAIL
    if (!skip_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    skip_wait = false;
    (void)skip_wait;
    const int m_a = (cur ^ 1) * WSTAGE_BYTES + w * 1024;
    if (kt == 0) wide4_ktile_a<true>(xa0, xa1, wa0, wa1, ag.aptr, m_a);
    else wide4_ktile_a<false>(xa0, xa1, wa0, wa1, ag.aptr, m_a);
#ifndef WIDE_X_NONEXT             /* timing knob: sources never advance (wrong results) */
    next_a();                   // K-tile s+2's A-row sources, while K-step 0's MFMAs drain
#endif
    {
      const int d = cur ? -WSTAGE_BYTES : WSTAGE_BYTES;
      wide4_ktile_b(wa1, in, m_a, wa0 + d, xa0 + d);
    }
#ifndef WIDE_X_NONEXT
    next_w(in);                 // ... and its weight-row sources
#endif
    {
      const int d = cur ? -WSTAGE_BYTES : WSTAGE_BYTES;      // the other stage becomes the current one
      xa0 += d; xa1 += d; wa0 += d; wa1 += d;
    }
    cur ^= 1;

    if (++kt == nk) {
      // ---------------------------------------------------------------- epilogue of `tile`, straight from registers.
      // Take step s+1's wait first (only K-tile s+1 is outstanding), so epilogue traffic never sits before it.
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // + the last MFMAs have left the pipe
      skip_wait = true;
      kt = 0;
      int tm, tn;
      supertile<4>(tile, tiles_m, tiles_n, tm, tn);
      tile += nc;
      const int m0 = tm * WBM + wr * 128 + l15;
      const int n0 = tn * WBN + wc * 160 + 4 * lq;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      // one token fragment (16 tokens x 160 channels) at a time
      ep_par ^= 1;
      const float* bl = (const float*)(smem + WBIAS_OFF + ep_par * (WBN * 4)) + wc * 160 + 4 * lq;   // bias[n0 + ...]
      const half_t* rbl = (const half_t*)(smem + WRB_OFF + ep_par * (2 * WRB_STRIP)) + wc * 160 + 4 * lq;
      // rows below rb_bound use the first strip, the others the second (the map changes at most once inside the tile)
      const unsigned rb_bound = rb_lds ? ((unsigned)(tm * WBM) / (unsigned)p.rb_d1 + 1u) * (unsigned)p.rb_d1 : 0u;
      auto epi = [&](int j, const float4_t (&e)[10]) {
        const long long m = m0 + j * 16;
        if (m >= p.M) return;
        if (MODE != LKGD_A_PLAIN || !p.geglu) {
          unsigned idx = 0;          // 32-bit row-map arithmetic: M < 2^24 is a launch condition of this kernel
          if (rbp && !rb_lds) idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) +
                                     (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          const int rb_sel = (unsigned)m < rb_bound ? 0 : WRB_STRIP / 2;
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            const int n = n0 + i * 16;
            if (n < p.N) {
              float4_t v = e[i];
              if (p.bias) v += *(const float4_t*)(bl + i * 16);
              if (rbp) {
                const half4_t rb = rb_lds ? *(const half4_t*)(rbl + rb_sel + i * 16)
                                          : *(const half4_t*)(rbp + (long long)idx * p.ldrb + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += (float)rb[x];
              }
              v *= p.s_acc;
              if (r1p) {
                half4_t r = *(const half4_t*)(r1p + m * p.ldr1 + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += p.r1 * (float)r[x];
              }
              if (r2p) {
                half4_t r = *(const half4_t*)(r2p + m * p.ldr2 + n);
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] += p.r2 * (float)r[x];
              }
              half4_t o;
#pragma unroll
              for (int x = 0; x < 4; ++x) o[x] = (half_t)v[x];
              WIDE_STORE_GUARD WIDE_ST((half4_t*)(outp + WIDE_OUT_ROW(m) * p.ldc + n), o);
            }
          }
        } else {
          // wave channels [0,80) = hidden, [80,160) = gate of output columns tn*160 + wc*80 + [0,80)
          const int oc0 = tn * 160 + wc * 80 + 4 * lq;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            float4_t hv = e[i], gv = e[i + 5];
            if (p.bias) {
              hv += *(const float4_t*)(bl + i * 16);
              gv += *(const float4_t*)(bl + 80 + i * 16);
            }
            const float2_t lo = __builtin_shufflevector(hv, hv, 0, 1) * gelu_erf2(__builtin_shufflevector(gv, gv, 0, 1));
            const float2_t hi = __builtin_shufflevector(hv, hv, 2, 3) * gelu_erf2(__builtin_shufflevector(gv, gv, 2, 3));
            const half4_t o = {(half_t)lo.x, (half_t)lo.y, (half_t)hi.x, (half_t)hi.y};
            WIDE_STORE_GUARD WIDE_ST((half4_t*)(outp + WIDE_OUT_ROW(m) * p.ldc + oc0 + i * 16), o);
          }
        }
      };
#define WIDE4_EPI(J)                                                                                               \
  {                                                                                                                \
    const float4_t e[10] = {wide4_read_acc<0, J>(), wide4_read_acc<1, J>(), wide4_read_acc<2, J>(), wide4_read_acc<3, J>(), \
                            wide4_read_acc<4, J>(), wide4_read_acc<5, J>(), wide4_read_acc<6, J>(), wide4_read_acc<7, J>(), \
                            wide4_read_acc<8, J>(), wide4_read_acc<9, J>()};                                       \
    epi(J, e);                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);  /* keep one token fragment's loads/stores from piling onto the next */   \
  }
      WIDE4_EPI(0) WIDE4_EPI(1) WIDE4_EPI(2) WIDE4_EPI(3) WIDE4_EPI(4) WIDE4_EPI(5) WIDE4_EPI(6) WIDE4_EPI(7)
#undef WIDE4_EPI
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's trailing loads must land before the LDS is released
}

extern "C" int lkgd_gemm_wide4_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lkgd_gemm_wide4_kernel<LKGD_A_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WLDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_wide4_kernel<LKGD_A_CONV3X3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WLDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lkgd_gemm_wide4_kernel<LKGD_A_TCONV3>, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  int tiles_m = (d->M + WBM - 1) / WBM, tiles_n = (d->N + WBN - 1) / WBN;
  long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  int grid = ntiles < cus ? (int)ntiles : cus;
  if (d->M >= (1 << 24)) return LKGD_E_SHAPE;            // float-reciprocal row decomposition (gemm_common.h)
  if (d->mode == LKGD_A_PLAIN)
    hipLaunchKernelGGL(lkgd_gemm_wide4_kernel<LKGD_A_PLAIN>, dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n);
  else if (d->mode == LKGD_A_CONV3X3)
    hipLaunchKernelGGL(lkgd_gemm_wide4_kernel<LKGD_A_CONV3X3>, dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n);
  else if (d->mode == LKGD_A_TCONV3)
    hipLaunchKernelGGL(lkgd_gemm_wide4_kernel<LKGD_A_TCONV3>, dim3(grid), dim3(WNT), WLDS, stream, *d, tiles_m, tiles_n);
  else
    return LKGD_E_MODE;
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
