// "Duo" MFMA GEMM for the short-K linears (plain A operand): 128 (tokens) x 320 (channels) output tiles, TWO workgroups of
// four waves per CU.
//
// Why: with K = 320 .. 1280 and 258k / 64k rows the 256x320 kernel (gemm_wide.hip) spends a third of its time in the
// epilogue: a CU turns 164 KB of accumulators into stores every 5-10 K-tiles, the chip absorbs stores at ~6-7 TB/s
// whatever their shape (tools/micro/store_bw.hip), and with one workgroup per CU the matrix pipe idles meanwhile (stores
// compiled out: 0.256 -> 0.170 ms on the 960 x 320 QKV projection of the 72x128 level; leaving them in flight longer, rows
// through LDS or staggered workgroups do not help - tools/micro/wide_knobs.sh).  Here the same wave tile (64 tokens x 160
// channels, 4 x 10 fragments of v_mfma_f32_16x16x32_f16, accumulators pinned in a[0:159]) runs in workgroups of 2 x 2
// waves, two of them resident per CU with nothing shared between them: one's epilogue overlaps the other's K-loop.
//
//   * BK = 32: a K-tile is ONE MFMA K-step.  A fragment (16 rows x 64 bytes) is one contiguous KiB of LDS, written by one
//     LDS-DMA wave instruction (lane -> row lane/4, 16-byte chunk lane%4 XOR (row/4)%4, so the fragment read
//     row*64 + (chunk ^ (row/4)%4)*16 is bank-conflict free); a stage is 8 token + 20 weight fragments = 28 KiB, two
//     stages + bias / row-bias strips = 61.5 KiB per workgroup;
//   * persistent workgroups, K-tile stream continuous across output tiles, XCD-cooperative tile order, register epilogue
//     (bias, row-indexed bias, scale, two residuals, GEGLU at interleave 80): as gemm_wide.hip;
//   * plain single-source A only (the convolutions have deep K and keep the 256x320 tiles).
#include "gemm_common.h"
#include "gemm_duo_ktile.inc"

#define DBM 128
#define DBN 320
#define DBK 32
#define DNT 256
#define DSTAGE ((DBM + DBN) * DBK * 2)       // 28 KiB
#define DBIAS_OFF (2 * DSTAGE)               // two 320-float bias strips (tile parity)
#define DRB_OFF (DBIAS_OFF + 2 * DBN * 4)    // row-bias strips: [tile parity][first | last row's table row][384 halfs]
#define DRB_STRIP 768
#define DLDS (DRB_OFF + 4 * DRB_STRIP)       // 61.5 KiB

struct DuoIn {                  // DMA sources of the NEXT K-tile besides the two A chunks
  unsigned oB[5];               // byte offsets of this lane's five weight chunks from wk
  const half_t* wk;             // weights + K offset (wave-uniform)
};

// barrier, fragment reads, the seven DMA loads of the next K-tile, 40 MFMAs (tools/gen_wide_asm.py)
template <bool FIRST>
__device__ __forceinline__ void duo_ktile(int xa, int wa, const half_t* const (&pA)[2], const DuoIn& in, int m_a) {
  half8_t x0, x1, x2, x3, w0, w1;
#define DUO_STMT(BODY)                                                                                              \
  asm volatile(BODY                                                                                                 \
               : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(w0), "=&v"(w1)                                   \
               : "v"(xa), "v"(wa), "v"(pA[0]), "v"(pA[1]), "v"(in.oB[0]), "v"(in.oB[1]), "v"(in.oB[2]),             \
                 "v"(in.oB[3]), "v"(in.oB[4]), "s"(in.wk), "s"(m_a)                                                 \
               : "memory", "scc", DUO_AGPR_CLOBBERS)
  if (FIRST) {
    DUO_STMT(DUO_KTILE_ASM_FIRST);
  } else {
    DUO_STMT(DUO_KTILE_ASM_NEXT);
  }
#undef DUO_STMT
}

// accumulator fragment (weight fragment i, token fragment j) out of the AGPRs; BASE = (4i + j) * 4
template <int BASE>
__device__ __forceinline__ float4_t duo_read_acc() {
  float a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\t"
               "v_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
               : "i"(BASE));
  return (float4_t){a, b, c, d};
}

template <int V> struct DuoIC { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void duo_static_for(F&& f, DuoIC<Is>...) { (f(DuoIC<Is>{}), ...); }
template <class F> __device__ __forceinline__ void duo_for10(F&& f) {
  duo_static_for(f, DuoIC<0>{}, DuoIC<1>{}, DuoIC<2>{}, DuoIC<3>{}, DuoIC<4>{}, DuoIC<5>{}, DuoIC<6>{}, DuoIC<7>{}, DuoIC<8>{}, DuoIC<9>{});
}
template <class F> __device__ __forceinline__ void duo_for5(F&& f) {
  duo_static_for(f, DuoIC<0>{}, DuoIC<1>{}, DuoIC<2>{}, DuoIC<3>{}, DuoIC<4>{});
}

__global__ __launch_bounds__(DNT, 2) __attribute__((amdgpu_num_vgpr(96))) void lkgd_gemm_duo_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n,
                                                                                                  int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;

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
  const int nk = p.K / DBK;
  const int total = (my_tiles > 0 ? my_tiles : 0) * nk;
  if (total <= 0) return;
  // The second workgroup of a CU (dispatch fills every CU of an XCD once before it doubles up: the upper half of the
  // per-XCD index) starts about half a tile late, so that the two epilogues of a CU do not coincide from the start.
  if (2 * c >= nc)
    for (int i = 0; i < stagger * nk; ++i) __builtin_amdgcn_s_sleep(8);      // 8 * 64 clocks each

  // ---- staging state: describes the K-tile whose loads are issued next.  Lane -> (row of the fragment, logical chunk)
  const int frow = lane >> 2;
  const int fch = (lane & 3) ^ ((lane >> 4) & 3);
  const half_t* aptr[2];
  DuoIn in;
  int st_tile = tile_begin - nc, st_kt = nk - 1, st_s = -1, st_par = 0, ep_par = 0;
  aptr[0] = aptr[1] = (const half_t*)p.a0;
#pragma unroll
  for (int i = 0; i < 5; ++i) in.oB[i] = 0;
  // Row-indexed bias with a row map that is piecewise constant over >= 128 rows: a tile's rows select at most two table
  // rows, those of its first and last token; both strips ride the LDS-DMA stream like the bias (see gemm_wide.hip)
  const bool rb_lds = p.rowbias && p.rb_d2 == 1 && p.rb_d1 >= DBM;
  const half_t* wbase = (const half_t*)p.w;
  asm volatile("" : "+s"(wbase));
  auto next_in = [&]() {
    if (st_s + 1 < total) {
      ++st_s;
      if (++st_kt == nk) {
        st_kt = 0;
        st_tile += nc;
        int tm, tn;
        supertile<4>(st_tile, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int m = tm * DBM + (w + 4 * i) * 16 + frow;
          m = m < p.M ? m : p.M - 1;      // clamped: rows past M are computed on a copy of the last row, never stored
          aptr[i] = (const half_t*)p.a0 + (long long)m * p.lda0 + fch * 8;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          int n = tn * DBN + (w + 4 * i) * 16 + frow;
          n = n < p.N ? n : p.N - 1;
          in.oB[i] = ((unsigned)n * (unsigned)p.K + fch * 8) * 2u;
        }
        st_par ^= 1;
        if (p.bias) {                     // 320 floats: dwords w*64 + lane, wave 0 also the last 64
          int n = tn * DBN + w * 64 + lane;
          n = n < p.N ? n : p.N - 1;
          __builtin_amdgcn_global_load_lds(GLB_PTR(p.bias + n), LDS_PTR(smem + DBIAS_OFF + st_par * (DBN * 4) + w * 256), 4, 0, 0);
          if (w == 0) {
            int n2 = tn * DBN + 256 + lane;
            n2 = n2 < p.N ? n2 : p.N - 1;
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.bias + n2), LDS_PTR(smem + DBIAS_OFF + st_par * (DBN * 4) + 1024), 4, 0, 0);
          }
        }
        if (rb_lds && w >= 1) {
          // waves 1-3 fetch dwords (w-1)*64 + lane of the two 160-dword strips (lanes past 160 re-read the last dword
          // into the strip's padding)
          const unsigned mf = (unsigned)(tm * DBM);
          unsigned ml = mf + DBM - 1;
          ml = ml < (unsigned)p.M ? ml : (unsigned)p.M - 1;
          const unsigned i0 = ((mf / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          const unsigned i1 = ((ml / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          int dw = (w - 1) * 64 + lane;
          int col = tn * DBN + 2 * (dw < 160 ? dw : 159);
          col = col < p.N - 2 ? col : p.N - 2;
          char* dst = smem + DRB_OFF + st_par * (2 * DRB_STRIP) + (w - 1) * 256;
          const half_t* rb = (const half_t*)p.rowbias;
          __builtin_amdgcn_global_load_lds(GLB_PTR(rb + (long long)i0 * p.ldrb + col), LDS_PTR(dst), 4, 0, 0);
          __builtin_amdgcn_global_load_lds(GLB_PTR(rb + (long long)i1 * p.ldrb + col), LDS_PTR(dst + DRB_STRIP), 4, 0, 0);
        }
      } else {
        aptr[0] += DBK;
        aptr[1] += DBK;
      }
    }
    in.wk = wbase + st_kt * DBK;
  };

  // fragment reads: token fragments wr*4 + j, weight fragments 8 + wc*10 + i of the CURRENT stage
  int xa, wa;
  {
    const int l15 = lane & 15, lq = lane >> 4;
    const int rsw = l15 * 64 + ((lq ^ ((l15 >> 2) & 3)) << 4);
    xa = wr * 4096 + rsw;
    wa = DBM * DBK * 2 + wc * 10240 + rsw;
  }

  // ---- K-tile 0 -> stage 0
  {
    next_in();
    char* sx = smem + w * 1024;
    glds16(aptr[0], sx);
    glds16(aptr[1], sx + 4096);
#pragma unroll
    for (int i = 0; i < 5; ++i) glds16((const char*)in.wk + in.oB[i], sx + DBM * DBK * 2 + 4096 * i);
  }
  next_in();                    // K-tile 1

  int cur = 0, kt = 0, tile = tile_begin;
  bool skip_wait = false;
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    // K-tile s must have landed (the only LDS-DMA batch in flight here); the body starts with the workgroup barrier,
    // issues K-tile s+1's loads into the other stage and computes K-tile s
    if (!skip_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    skip_wait = false;
    const int m_a = (cur ^ 1) * DSTAGE + w * 1024;
    if (kt == 0) duo_ktile<true>(xa, wa, aptr, in, m_a);
    else duo_ktile<false>(xa, wa, aptr, in, m_a);
    next_in();                  // K-tile s+2's sources, for the next body
    {
      const int d = cur ? -DSTAGE : DSTAGE;
      xa += d; wa += d;
    }
    cur ^= 1;

    if (++kt == nk) {
      // ---------------------------------------------------------------- epilogue of `tile`, straight from registers.
      // Take step s+1's wait first (only K-tile s+1 is outstanding), so epilogue traffic never sits before it.
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // + the last MFMAs have left the pipe
      skip_wait = true;
      kt = 0;
      int lane_e = lane;             // opaque copy: everything per-lane below is recomputed here, once per tile, instead of
      asm volatile("" : "+v"(lane_e));   // being hoisted out of the persistent loop into registers the K-tile body needs
      const int l15 = lane_e & 15, lq = lane_e >> 4;
      int tm, tn;
      supertile<4>(tile, tiles_m, tiles_n, tm, tn);
      tile += nc;
      const int m0 = tm * DBM + wr * 64 + l15;
      const int n0 = tn * DBN + wc * 160 + 4 * lq;
      const half_t* rbp = (const half_t*)p.rowbias;
      const half_t* r1p = (const half_t*)p.res1;
      const half_t* r2p = (const half_t*)p.res2;
      half_t* outp = (half_t*)p.out;
      ep_par ^= 1;
      const float* bl = (const float*)(smem + DBIAS_OFF + ep_par * (DBN * 4)) + wc * 160 + 4 * lq;   // bias[n0 + ...]
      const half_t* rbl = (const half_t*)(smem + DRB_OFF + ep_par * (2 * DRB_STRIP)) + wc * 160 + 4 * lq;
      // rows below rb_bound use the first strip, the others the second (the map changes at most once inside the tile)
      const unsigned rb_bound = rb_lds ? ((unsigned)(tm * DBM) / (unsigned)p.rb_d1 + 1u) * (unsigned)p.rb_d1 : 0u;
      // one token fragment (16 tokens x 160 channels) at a time; accumulator fragments are read where they are used
      auto epi = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const long long m = m0 + j * 16;
        if (m >= p.M) return;
        if (!p.geglu) {
          unsigned idx = 0;          // 32-bit row-map arithmetic: M < 2^24 is a launch condition of this kernel
          if (rbp && !rb_lds) idx = (((unsigned)m / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + ((unsigned)m % (unsigned)p.rb_d2) +
                                     (unsigned)p.rb_c0) % (unsigned)p.rb_md;
          const int rb_sel = (unsigned)m < rb_bound ? 0 : DRB_STRIP / 2;
          duo_for10([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int n = n0 + i * 16;
            float4_t v = duo_read_acc<(4 * i + j) * 4>();
            if (n < p.N) {
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
              *(half4_t*)(outp + m * p.ldc + n) = o;
            }
          });
        } else {
          // wave channels [0,80) = hidden, [80,160) = gate of output columns tn*160 + wc*80 + [0,80)
          const int oc0 = tn * 160 + wc * 80 + 4 * lq;
          duo_for5([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            float4_t hv = duo_read_acc<(4 * i + j) * 4>(), gv = duo_read_acc<(4 * (i + 5) + j) * 4>();
            if (p.bias) {
              hv += *(const float4_t*)(bl + i * 16);
              gv += *(const float4_t*)(bl + 80 + i * 16);
            }
            const float2_t lo = __builtin_shufflevector(hv, hv, 0, 1) * gelu_erf2(__builtin_shufflevector(gv, gv, 0, 1));
            const float2_t hi = __builtin_shufflevector(hv, hv, 2, 3) * gelu_erf2(__builtin_shufflevector(gv, gv, 2, 3));
            const half4_t o = {(half_t)lo.x, (half_t)lo.y, (half_t)hi.x, (half_t)hi.y};
            *(half4_t*)(outp + m * p.ldc + oc0 + i * 16) = o;
          });
        }
      };
#define DUO_EPI(J)                                                                                                 \
  {                                                                                                                \
    epi(DuoIC<J>{});                                                                                               \
    __builtin_amdgcn_sched_barrier(0);  /* keep one token fragment's loads/stores from piling onto the next */   \
  }
      DUO_EPI(0) DUO_EPI(1) DUO_EPI(2) DUO_EPI(3)
#undef DUO_EPI
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's trailing loads must land before the LDS is released
}

static int duo_stagger = 8;       // start offset of a CU's second workgroup, in 512-clock sleeps per K-tile of a tile
extern "C" void lkgd_debug_set_duo_stagger(int v) { duo_stagger = v < 0 ? 0 : v; }

// d: plain mode, single-source A, K % 32 == 0, geglu 0 or 80 (checked by the dispatcher, re-checked here)
extern "C" int lkgd_gemm_duo_launch(const lkgd_gemm_desc* d, hipStream_t stream, int cus) {
  LKGD_DEVICE_ONCE_BEGIN
    if (hipFuncSetAttribute((const void*)lkgd_gemm_duo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DLDS) != hipSuccess)
      return LKGD_E_LAUNCH;
  LKGD_DEVICE_ONCE_END
  if (d->mode != LKGD_A_PLAIN || d->csplit < d->K || d->K % DBK || (d->geglu != 0 && d->geglu != 80)) return LKGD_E_SHAPE;
  if (d->M < 1 || d->M >= (1 << 24) || d->N < 2) return LKGD_E_SHAPE;
  if (d->geglu && d->N % DBN) return LKGD_E_SHAPE;
  const int tiles_m = (d->M + DBM - 1) / DBM, tiles_n = (d->N + DBN - 1) / DBN;
  const long long ntiles = (long long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffLL) return LKGD_E_SHAPE;
  const int slots = 2 * cus;
  const int grid = ntiles < slots ? (int)ntiles : slots;
  hipLaunchKernelGGL(lkgd_gemm_duo_kernel, dim3(grid), dim3(DNT), DLDS, stream, *d, tiles_m, tiles_n, duo_stagger);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
