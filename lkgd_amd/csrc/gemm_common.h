// Shared pieces of the MFMA GEMM kernels: the implicit-convolution A-operand gather and the epilogue row map.
#pragma once
#include "common.h"

#define BK 64

// K order of the 3x3 implicit convolution (weights packed by lkgd_amd/packing.py::pack_conv3x3):
//   k = ((ky * (Cin/64) + c/64) * 3 + kx) * 64 + c % 64
// i.e. for one kernel row ky and one 64-channel chunk the three horizontal taps are consecutive K-tiles.  They read the same
// source cache lines shifted by one pixel, so the second and third hit the XCD's L2 (with tap-major order, k = tap * Cin + c,
// a tap's lines came back Cin/64 K-tiles later, after ~10 MB of other traffic through a 4 MB L2: the A panel was fetched
// from the fabric once per tap, profiles/r02_pmc_hbm_traffic.txt).  Every K-tile is its own gather segment.
__device__ __forceinline__ void conv_k_decode(int k0, int Cin, int& ky, int& kx, int& c) {
  const int t = k0 >> 6;                 // K-tile index (BK = 64)
  const int nc = Cin >> 6;
  const int q = t / 3;                   // ky * nc + chunk   (wave-uniform scalar divisions)
  kx = t - q * 3;
  ky = q / nc;
  c = (q - ky * nc) << 6;
}

struct ARow {
  long long base;  // mode-specific row base (token index of the source image / row)
  int y, x;        // conv: top-left of the 3x3 window in the (virtual) source grid; tconv: frame index in y
  int valid;
};

// MODE_ >= 0 fixes the A-operand mode at compile time (kernels instantiated per mode); -1 reads it from the descriptor
template <int MODE_ = -1>
__device__ __forceinline__ const half_t* a_source(const lkgd_gemm_desc& p, const ARow& r, int k0, int chunk) {
  const half_t* zero = (const half_t*)p.zeros;
  const int mode = MODE_ >= 0 ? MODE_ : p.mode;
  if (!r.valid) return zero;
  if (mode == LKGD_A_PLAIN) {
    int k = k0 + chunk * 8;
    if (k < p.csplit) return (const half_t*)p.a0 + r.base * p.lda0 + k;
    return (const half_t*)p.a1 + r.base * p.lda1 + (k - p.csplit);
  }
  if (mode == LKGD_A_CONV3X3) {
    int ky, kx, c;
    conv_k_decode(k0, p.Cin, ky, kx, c);
    c += chunk * 8;
    int vy = r.y + ky, vx = r.x + kx;
    int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    if ((unsigned)vy >= (unsigned)Hv || (unsigned)vx >= (unsigned)Wv) return zero;
    long long row = r.base + (long long)(vy >> p.ups) * p.Win + (vx >> p.ups);
    if (c < p.csplit) return (const half_t*)p.a0 + row * p.lda0 + c;
    return (const half_t*)p.a1 + row * p.lda1 + (c - p.csplit);
  }
  if (mode == LKGD_A_TCONV3) {
    int tap = k0 / p.Cin;
    int c = k0 - tap * p.Cin + chunk * 8;
    int f = r.y + tap - 1;
    if ((unsigned)f >= (unsigned)p.F) return zero;
    long long row = r.base + (long long)f * p.HW;
    return (const half_t*)p.a0 + row * p.lda0 + c;
  }
  // LKGD_A_CONV3X3_C8: one 16-byte chunk (8 channels) per tap
  int tap = (k0 >> 3) + chunk;
  if (tap >= 9) return zero;
  int ky = tap / 3, kx = tap - ky * 3;
  int vy = r.y + ky, vx = r.x + kx;
  if ((unsigned)vy >= (unsigned)p.Hin || (unsigned)vx >= (unsigned)p.Win) return zero;
  long long row = r.base + (long long)vy * p.Win + vx;
  return (const half_t*)p.a0 + row * p.lda0;
}


// per-thread decomposition of output row m into the gather state of its mode
template <int MODE_ = -1>
__device__ __forceinline__ ARow a_row(const lkgd_gemm_desc& p, int m) {
  const int mode = MODE_ >= 0 ? MODE_ : p.mode;
  ARow r;
  r.valid = m < p.M;
  r.base = m; r.y = 0; r.x = 0;
  if (r.valid) {
    if (mode == LKGD_A_CONV3X3 || mode == LKGD_A_CONV3X3_C8) {
      int hw = p.Hout * p.Wout;
      int n = m / hw, rem = m - n * hw;
      int y = rem / p.Wout, x = rem - y * p.Wout;
      r.base = (long long)n * p.Hin * p.Win;
      r.y = y * p.stride - 1 + p.pad_off;
      r.x = x * p.stride - 1 + p.pad_off;
    } else if (mode == LKGD_A_TCONV3) {
      int bf = m / p.HW;                 // b*Floc + fl
      int b = bf / p.Floc;
      r.y = bf - b * p.Floc + p.f_off;   // global frame
      r.base = (long long)b * p.F * p.HW + (m - (long long)bf * p.HW);   // + f*HW added per tap
    }
  }
  return r;
}

// ---- segment-wise A gather --------------------------------------------------------------------------------------
// K is walked in BK chunks; consecutive chunks stay inside one "segment" = (3x3 / temporal tap, source tensor) for
// Cin/64 (or csplit/64) chunks.  All the per-lane address work (halo test, upsample shift, frame shift, concat source
// select) is done ONCE per segment; inside a segment a chunk's source is `ptr + (k0 - seg_k0)` (zero page rows stay).
template <int NR>
struct AGather {
  const half_t* ptr[NR];
  ARow row[NR];
  unsigned zmask;        // bit i: row i reads the zero page in this segment
  int seg_k0, seg_end;
};

template <int NR, int MODE_ = -1>
__device__ __forceinline__ void a_segment(const lkgd_gemm_desc& p, AGather<NR>& g, int k0, int schunk) {
  const half_t* zero = (const half_t*)p.zeros;
  const int mode = MODE_ >= 0 ? MODE_ : p.mode;
  g.zmask = 0;
  if (mode == LKGD_A_PLAIN) {
    const bool s1 = k0 >= p.csplit;
    g.seg_k0 = s1 ? p.csplit : 0;
    g.seg_end = s1 ? p.K : (p.csplit < p.K ? p.csplit : p.K);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const ARow& r = g.row[i];
      const half_t* q = s1 ? (const half_t*)p.a1 + r.base * p.lda1 : (const half_t*)p.a0 + r.base * p.lda0;
      g.ptr[i] = r.valid ? q + schunk * 8 : zero;
      g.zmask |= (r.valid ? 0u : 1u) << i;
    }
  } else if (mode == LKGD_A_CONV3X3) {
    int ky, kx, c;
    conv_k_decode(k0, p.Cin, ky, kx, c);
    const bool s1 = c >= p.csplit;
    g.seg_k0 = k0;                       // one K-tile per segment: the tap changes with every K-tile
    g.seg_end = k0 + BK;
    const int cs = s1 ? c - p.csplit : c;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const ARow& r = g.row[i];
      const int vy = r.y + ky, vx = r.x + kx;
      const bool ok = r.valid && (unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv;
      const long long rowi = r.base + (long long)(vy >> p.ups) * p.Win + (vx >> p.ups);
      const half_t* q = s1 ? (const half_t*)p.a1 + rowi * p.lda1 : (const half_t*)p.a0 + rowi * p.lda0;
      g.ptr[i] = ok ? q + cs + schunk * 8 : zero;
      g.zmask |= 1u << i;                // pointer is final: no in-segment offset
    }
  } else if (mode == LKGD_A_TCONV3) {
    const int tap = k0 / p.Cin;
    g.seg_k0 = tap * p.Cin;
    g.seg_end = g.seg_k0 + p.Cin;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const ARow& r = g.row[i];
      const int f = r.y + tap - 1;
      const bool ok = r.valid && (unsigned)f < (unsigned)p.F;
      const long long rowi = r.base + (long long)f * p.HW;
      g.ptr[i] = ok ? (const half_t*)p.a0 + rowi * p.lda0 + schunk * 8 : zero;
      g.zmask |= (ok ? 0u : 1u) << i;
    }
  } else {   // LKGD_A_CONV3X3_C8: one tap per 16-byte chunk - every K-tile is its own segment
    g.seg_k0 = k0;
    g.seg_end = k0 + BK;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      g.ptr[i] = a_source<MODE_>(p, g.row[i], k0, schunk);
      g.zmask |= 1u << i;     // pointer is final: no in-segment offset
    }
  }
}

// source of row i's chunk for K-tile k0 (k0 must lie in the current segment)
template <int NR>
__device__ __forceinline__ const half_t* a_chunk(const AGather<NR>& g, int i, int k0) {
  const int off = ((g.zmask >> i) & 1u) ? 0 : (k0 - g.seg_k0);
  return g.ptr[i] + off;
}

// ---- lean gather -----------------------------------------------------------------------------------------------
// For kernels whose accumulators leave few registers (gemm_wide.hip): per A row a compact descriptor
// (2 registers) and the current segment's source pointer (2 registers).  The per-segment update has no divisions; the
// per-tile decomposition of the row index uses a float-reciprocal division (exact for M < 2^24), so deriving a tile's
// rows does not spike the register pressure inside a K-loop the way a_row()'s integer divisions do.
struct RowD {
  int base;     // conv: n*Hin*Win; tconv: b*F*HW + pixel; plain: m.   -1 = row past M (reads the zero page)
  int yx;       // conv: (vy0 & 0xffff) | (vx0 << 16), top-left of the 3x3 window in the virtual source grid; tconv: frame
};

__device__ __forceinline__ int fast_div(int m, int d, float rcp_d) {   // floor(m / d) for 0 <= m < 2^24, d > 0
  int q = (int)((float)m * rcp_d);
  int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

template <int MODE>
__device__ __forceinline__ void lean_rcps(const lkgd_gemm_desc& p, float& rcp0, float& rcp1) {
  rcp0 = 1.f; rcp1 = 1.f;
  if (MODE == LKGD_A_CONV3X3) { rcp0 = 1.0f / (float)(p.Hout * p.Wout); rcp1 = 1.0f / (float)p.Wout; }
  if (MODE == LKGD_A_TCONV3) { rcp0 = 1.0f / (float)p.HW; rcp1 = 1.0f / (float)p.Floc; }
}

template <int MODE>
__device__ __forceinline__ RowD lean_row(const lkgd_gemm_desc& p, int m, float rcp0, float rcp1) {
  RowD r;
  r.base = -1; r.yx = 0;
  if (m < p.M) {
    if (MODE == LKGD_A_CONV3X3) {
      const int hw = p.Hout * p.Wout;
      const int n = fast_div(m, hw, rcp0), rem = m - n * hw;
      const int y = fast_div(rem, p.Wout, rcp1), x = rem - y * p.Wout;
      r.base = n * p.Hin * p.Win;
      r.yx = ((y * p.stride - 1 + p.pad_off) & 0xffff) | ((x * p.stride - 1 + p.pad_off) << 16);
    } else if (MODE == LKGD_A_TCONV3) {
      const int bf = fast_div(m, p.HW, rcp0);               // b*Floc + fl
      const int b = fast_div(bf, p.Floc, rcp1);
      r.yx = bf - b * p.Floc + p.f_off;                      // global frame
      r.base = b * p.F * p.HW + (m - bf * p.HW);             // + f*HW added per tap
    } else {
      r.base = m;
    }
  }
  return r;
}

template <int NR>
struct LeanGather {
  RowD rd[NR];
  const half_t* aptr[NR];       // source of this thread's A rows in the current segment (chunk offset included)
  unsigned zmask;               // bit i: row i reads the zero page in this segment
  int seg_k0, seg_end;          // K range of the current segment (wave-uniform)
  int ky, kx, cc;               // 3x3 conv: tap / first channel of the current segment (wave-uniform), stepped without divisions
};

// source pointers of the rows for the segment containing k0
template <int MODE, int NR>
__device__ __forceinline__ void lean_segment(const lkgd_gemm_desc& p, LeanGather<NR>& st, int k0, int schunk) {
  const half_t* zero = (const half_t*)p.zeros;
  st.zmask = 0;
  if (MODE == LKGD_A_PLAIN) {
    const bool s1 = k0 >= p.csplit;
    st.seg_k0 = s1 ? p.csplit : 0;
    st.seg_end = s1 ? p.K : (p.csplit < p.K ? p.csplit : p.K);
    const half_t* src = (s1 ? (const half_t*)p.a1 : (const half_t*)p.a0) + schunk * 8;
    const unsigned ld = s1 ? p.lda1 : p.lda0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = st.rd[i].base >= 0;
      st.aptr[i] = ok ? src + (unsigned long long)(unsigned)st.rd[i].base * ld : zero;
      st.zmask |= (ok ? 0u : 1u) << i;
    }
  } else if (MODE == LKGD_A_CONV3X3) {
    // consecutive K-tiles step (kx, chunk, ky) like an odometer; only a tile's (or K slice's) first K-tile is decoded by division
    if (k0 != 0 && k0 == st.seg_end) {
      if (++st.kx == 3) {
        st.kx = 0;
        st.cc += BK;
        if (st.cc == p.Cin) { st.cc = 0; ++st.ky; }
      }
    } else {
      conv_k_decode(k0, p.Cin, st.ky, st.kx, st.cc);       // wave-uniform (scalar) divisions
    }
    const int ky = st.ky, kx = st.kx, cc = st.cc;
    const bool s1 = cc >= p.csplit;
    st.seg_k0 = k0;                                        // one K-tile per segment
    st.seg_end = k0 + BK;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const half_t* src = (s1 ? (const half_t*)p.a1 + (cc - p.csplit) : (const half_t*)p.a0 + cc) + schunk * 8;
    const unsigned ld = s1 ? p.lda1 : p.lda0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int vy = (int)(short)(st.rd[i].yx & 0xffff) + ky, vx = (st.rd[i].yx >> 16) + kx;
      const bool ok = st.rd[i].base >= 0 && (unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv;
      const unsigned row = (unsigned)(st.rd[i].base + (vy >> p.ups) * p.Win + (vx >> p.ups));
      st.aptr[i] = ok ? src + (unsigned long long)row * ld : zero;
      st.zmask |= (ok ? 0u : 1u) << i;
    }
  } else {   // LKGD_A_TCONV3
    const int tap = k0 / p.Cin;
    st.seg_k0 = tap * p.Cin;
    st.seg_end = st.seg_k0 + p.Cin;
    const half_t* src = (const half_t*)p.a0 + schunk * 8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int f = st.rd[i].yx + tap - 1;
      const bool ok = st.rd[i].base >= 0 && (unsigned)f < (unsigned)p.F;
      const unsigned row = (unsigned)(st.rd[i].base + f * p.HW);
      st.aptr[i] = ok ? src + (unsigned long long)row * (unsigned)p.lda0 : zero;
      st.zmask |= (ok ? 0u : 1u) << i;
    }
  }
}

// source of row i's chunk for K-tile k0 (k0 must lie in the current segment)
template <int NR>
__device__ __forceinline__ const half_t* lean_chunk(const LeanGather<NR>& st, int i, int k0) {
  return st.aptr[i] + (((st.zmask >> i) & 1u) ? 0 : (k0 - st.seg_k0));
}

// Linear tile index -> (tm, tn) for the persistent kernels.  Tiles are ordered in column groups of GW n-tiles, m-major
// inside a group, so the ~32 tiles an XCD works on at any time form an (32/GW) x GW block: a weight K-slice is shared by
// 32/GW CUs and an A K-slice by GW (both through that XCD's L2), and the group's weight strip stays L2-resident while
// the XCD walks down the rows.  With n fastest over ALL n-tiles (20 - 40 of them for the GEGLU / QKV projections) every
// m-row re-streamed the whole weight matrix: 996 MB fetched per launch for 89 MB of operands
// (profiles/r01_pmc_hbm_traffic_pp.txt).
template <int GW>
__device__ __forceinline__ void supertile(int tile, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per_group = tiles_m * GW;
  const int g = tile / per_group;
  const int r = tile - g * per_group;
  const int n_first = g * GW;
  const int width = tiles_n - n_first < GW ? tiles_n - n_first : GW;
  tm = r / width;
  tn = n_first + (r - tm * width);
}

// XCD-aware, bijective block -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), so each residue class
// gets a contiguous range of tiles; inside the range n is fastest (the A tile is reused from that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int xcd = bid & 7, slot = bid >> 3;
  int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// shared epilogue on an fp32 C tile staged in LDS (row pitch BN_ floats): bias, row-indexed bias, scale, two residuals
// or GEGLU; NT threads, 8-byte row-contiguous stores
template <int BM_, int BN_, int NT>
__device__ __forceinline__ void gemm_epilogue(const lkgd_gemm_desc& p, const float* ct, int t, int m0, int n0,
                                              int tn) {
  const half_t* rbp = (const half_t*)p.rowbias;
  const half_t* r1p = (const half_t*)p.res1;
  const half_t* r2p = (const half_t*)p.res2;
  half_t* outp = (half_t*)p.out;
  if (!p.geglu) {
    constexpr int TPR = BN_ / 4;            // threads per row
    constexpr int RPP = NT / TPR;           // rows per pass
    const int col = (t % TPR) * 4;
    const int gcol = n0 + col;
    if (gcol < p.N) {
      float4_t bias = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias = *(const float4_t*)(p.bias + gcol);
      // rows in batches of EB: a batch's residual / row-bias loads are all issued before its arithmetic (one exposed memory
      // round trip per batch instead of one per row - with one workgroup on a CU nothing else covers them)
      constexpr int NIT = BM_ / RPP;
      constexpr int EB = NIT < 8 ? NIT : 8;
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += EB) {
        half4_t rbv[EB], r1v[EB], r2v[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const half4_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
          rbv[u] = z; r1v[u] = z; r2v[u] = z;
          const long long m = m0 + (t / TPR) + RPP * (it0 + u);
          if (m < p.M) {
            if (rbp) {
              // M is an int32: 32-bit unsigned row-map arithmetic (a 64-bit division is a few hundred instructions)
              const unsigned mu = (unsigned)m;
              const unsigned idx = ((mu / (unsigned)p.rb_d1) * (unsigned)p.rb_m1 + (mu % (unsigned)p.rb_d2) +
                                    (unsigned)p.rb_c0) % (unsigned)p.rb_md;
              rbv[u] = *(const half4_t*)(rbp + (long long)idx * p.ldrb + gcol);
            }
            if (r1p) r1v[u] = *(const half4_t*)(r1p + m * p.ldr1 + gcol);
            if (r2p) r2v[u] = *(const half4_t*)(r2p + m * p.ldr2 + gcol);
          }
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const int row = (t / TPR) + RPP * (it0 + u);
          const long long m = m0 + row;
          if (m >= p.M) break;
          float4_t v = *(const float4_t*)(ct + row * BN_ + col);
          v += bias;
          if (rbp) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rbv[u][e];
          }
          v *= p.s_acc;
          if (r1p) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += p.r1 * (float)r1v[u][e];
          }
          if (r2p) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += p.r2 * (float)r2v[u][e];
          }
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
          *(half4_t*)(outp + m * p.ldc + gcol) = o;
        }
      }
    }
  } else {
    // tile columns: [0,32) hidden 0-31 | [32,64) gate 0-31 | [64,96) hidden 32-63 | [96,128) gate 32-63
    static_assert(BN_ == 128, "GEGLU tile interleave is 32 hidden | 32 gate | 32 hidden | 32 gate");
    constexpr int RPP = NT / 16;
    const int oc = (t & 15) * 4;                 // output column inside the tile's 64
    const int col = (oc >> 5) * 64 + (oc & 31);  // its hidden column in the tile; gate = +32
    const int ocol = tn * 64 + oc;
    float4_t bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      bh = *(const float4_t*)(p.bias + n0 + col);
      bg = *(const float4_t*)(p.bias + n0 + 32 + col);
    }
#pragma unroll 4
    for (int it = 0; it < BM_ / RPP; ++it) {
      int row = (t >> 4) + RPP * it;
      long long m = m0 + row;
      if (m >= p.M) break;
      float4_t hv = *(const float4_t*)(ct + row * BN_ + col) + bh;
      float4_t gv = *(const float4_t*)(ct + row * BN_ + 32 + col) + bg;
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
      *(half4_t*)(outp + m * p.ldc + ocol) = o;
    }
  }
}
