// MFMA fp16 GEMM with implicit-convolution A operand and fused epilogue (include/lkgd_hip.h section 1).
//
// Design (gfx950 / CDNA4):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 v_mfma_f32_32x32x16_f16,
//     64 fp32 accumulator VGPRs), BK = 64, two LDS stages of 32 KiB -> 2 workgroups per CU.
//   * both operands are K-contiguous ([M][K] tokens, [N][K] weights), so A and B fragments are one ds_read_b128 each.
//   * staging is LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear, the bank-conflict swizzle
//     (16-byte chunk ^= (row>>1)&7 inside each 128-byte row) is applied to the per-lane SOURCE address and to the
//     fragment read.  The per-lane source address is also what makes the A operand an implicit convolution: every
//     lane computes where "its" 16 bytes live (3x3 tap with halo, stride-2, nearest-2x upsample, frame shift,
//     two-tensor channel concat) and out-of-image lanes read a zero page.
//   * one barrier per K-tile (stage t+1 is issued before the MFMAs of tile t).
//   * epilogue: accumulators -> LDS (fp32, reusing the staging buffers) -> row-contiguous 8-byte stores with bias,
//     row-indexed bias (time embedding / positional / cross-attention), GEGLU, two scaled residuals (AlphaBlender).
//   * blockIdx -> tile map is XCD-aware (each XCD's L2 sees a contiguous range of tiles, n fastest).
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)  // A 16 KiB + B 16 KiB
#define GEMM_LDS (2 * STAGE_BYTES)     // 64 KiB (also holds the fp32 C tile in the epilogue)

struct ARow {
  long long base;  // mode-specific row base (token index of the source image / row)
  int y, x;        // conv: top-left of the 3x3 window in the (virtual) source grid; tconv: frame index in y
  int valid;
};

__device__ __forceinline__ const half_t* a_source(const lkgd_gemm_desc& p, const ARow& r, int k0, int chunk) {
  const half_t* zero = (const half_t*)p.zeros;
  if (!r.valid) return zero;
  if (p.mode == LKGD_A_PLAIN) {
    int k = k0 + chunk * 8;
    if (k < p.csplit) return (const half_t*)p.a0 + r.base * p.lda0 + k;
    return (const half_t*)p.a1 + r.base * p.lda1 + (k - p.csplit);
  }
  if (p.mode == LKGD_A_CONV3X3) {
    int tap = k0 / p.Cin;
    int c = k0 - tap * p.Cin + chunk * 8;
    int ky = tap / 3, kx = tap - ky * 3;
    int vy = r.y + ky, vx = r.x + kx;
    int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    if ((unsigned)vy >= (unsigned)Hv || (unsigned)vx >= (unsigned)Wv) return zero;
    long long row = r.base + (long long)(vy >> p.ups) * p.Win + (vx >> p.ups);
    if (c < p.csplit) return (const half_t*)p.a0 + row * p.lda0 + c;
    return (const half_t*)p.a1 + row * p.lda1 + (c - p.csplit);
  }
  if (p.mode == LKGD_A_TCONV3) {
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

__global__ __launch_bounds__(256, 2) void lkgd_gemm_kernel(const lkgd_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 1, wc = w & 1;

  // ---- XCD-aware, bijective block -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), so give
  //      each residue class a contiguous range of tiles; inside the range n is fastest (A tile reused from L2).
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, slot = bid >> 3;
    int q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread staging rows: thread t fills LDS slots t + 256*i (slot = row*8 + chunk), i = 0..3
  const int srow = t >> 3;                               // + 32*i
  const int schunk = (t & 7) ^ ((t >> 4) & 7);           // logical 16-byte chunk this thread fetches (swizzled)
  ARow ar[4];
  const half_t* brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 32 * i;
    ARow r;
    r.valid = m < p.M;
    r.base = m; r.y = 0; r.x = 0;
    if (r.valid) {
      if (p.mode == LKGD_A_CONV3X3 || p.mode == LKGD_A_CONV3X3_C8) {
        int hw = p.Hout * p.Wout;
        int n = m / hw, rem = m - n * hw;
        int y = rem / p.Wout, x = rem - y * p.Wout;
        r.base = (long long)n * p.Hin * p.Win;
        r.y = y * p.stride - 1;
        r.x = x * p.stride - 1;
      } else if (p.mode == LKGD_A_TCONV3) {
        int bf = m / p.HW;                 // b*Floc + fl
        int b = bf / p.Floc;
        r.y = bf - b * p.Floc + p.f_off;   // global frame
        r.base = (long long)b * p.F * p.HW + (m - (long long)bf * p.HW);   // + f*HW added per tap
      }
    }
    ar[i] = r;
    int n = n0 + srow + 32 * i;
    brow[i] = n < p.N ? (const half_t*)p.w + (long long)n * p.K + schunk * 8 : (const half_t*)p.zeros;
  }
  const int b_adv = 1;  // brow advances by BK per K-tile unless it is the zero page
  (void)b_adv;

  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + BM * BK * 2;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_source(p, ar[i], k0, schunk), sa + (w * 64 + 256 * i) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n = n0 + srow + 32 * i;
      const half_t* src = n < p.N ? brow[i] + k0 : (const half_t*)p.zeros;
      glds16(src, sb + (w * 64 + 256 * i) * 16);
    }
  };

  float16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses (bytes inside a stage): row*128 + ((ks*2 + h) ^ ((row>>1)&7))*16
  const int h = lane >> 5;
  int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = wr * 64 + i * 32 + (lane & 31);
    int rb = wc * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
    b_off[i] = BM * BK * 2 + rb * 128; b_sw[i] = (rb >> 1) & 7;
  }

  const int nk = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8_t af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *(const half8_t*)(sbase + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
        bf[i] = *(const half8_t*)(sbase + b_off[i] + (((ks * 2 + h) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS fp32 tile [128][128] -> row-contiguous stores
  float* ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        int col = wc * 64 + j * 32 + (lane & 31);
        ct[row * BN + col] = acc[i][j][r];
      }
  __syncthreads();

  const half_t* rbp = (const half_t*)p.rowbias;
  const half_t* r1p = (const half_t*)p.res1;
  const half_t* r2p = (const half_t*)p.res2;
  half_t* outp = (half_t*)p.out;

  if (!p.geglu) {
    const int col = (t & 31) * 4;
    const int gcol = n0 + col;
    if (gcol < p.N) {
      float4_t bias = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias = *(const float4_t*)(p.bias + gcol);
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        int row = (t >> 5) + 8 * it;
        long long m = m0 + row;
        if (m >= p.M) break;
        float4_t v = *(const float4_t*)(ct + row * BN + col);
        v += bias;
        if (rbp) {
          long long idx = ((m / p.rb_d1) * p.rb_m1 + (m % p.rb_d2) + p.rb_c0) % p.rb_md;
          half4_t rb = *(const half4_t*)(rbp + idx * p.ldrb + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
        }
        v *= p.s_acc;
        if (r1p) {
          half4_t r = *(const half4_t*)(r1p + m * p.ldr1 + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += p.r1 * (float)r[e];
        }
        if (r2p) {
          half4_t r = *(const half4_t*)(r2p + m * p.ldr2 + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += p.r2 * (float)r[e];
        }
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *(half4_t*)(outp + m * p.ldc + gcol) = o;
      }
    }
  } else {
    // tile columns [0,64) = hidden, [64,128) = gate of output columns tn*64 + [0,64)
    const int col = (t & 15) * 4;
    const int ocol = tn * 64 + col;
    float4_t bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      bh = *(const float4_t*)(p.bias + n0 + col);
      bg = *(const float4_t*)(p.bias + n0 + 64 + col);
    }
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      int row = (t >> 4) + 16 * it;
      long long m = m0 + row;
      if (m >= p.M) break;
      float4_t hv = *(const float4_t*)(ct + row * BN + col) + bh;
      float4_t gv = *(const float4_t*)(ct + row * BN + 64 + col) + bg;
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
      *(half4_t*)(outp + m * p.ldc + ocol) = o;
    }
  }
}

static int check_desc(const lkgd_gemm_desc* d) {
  if (!d || !d->a0 || !d->w || !d->out || !d->zeros) return LKGD_E_NULL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return LKGD_E_SHAPE;
  if (d->K % BK) return LKGD_E_SHAPE;
  if (d->N % 4) return LKGD_E_SHAPE;
  if (d->geglu && (d->N % 128 || d->rowbias || d->res1 || d->res2)) return LKGD_E_SHAPE;
  if (d->ldc % 4 || d->lda0 % 8) return LKGD_E_ALIGN;
  if (!aligned16(d->a0) || !aligned16(d->w) || !aligned16(d->zeros) || ((uintptr_t)d->out & 7)) return LKGD_E_ALIGN;
  if (d->a1 && (!aligned16(d->a1) || d->lda1 % 8)) return LKGD_E_ALIGN;
  if (d->bias && !aligned16(d->bias)) return LKGD_E_ALIGN;
  if (d->rowbias && (d->ldrb % 4 || d->rb_d1 <= 0 || d->rb_d2 <= 0 || d->rb_md <= 0 || d->rb_c0 < 0)) return LKGD_E_SHAPE;
  if (d->res1 && d->ldr1 % 4) return LKGD_E_ALIGN;
  if (d->res2 && d->ldr2 % 4) return LKGD_E_ALIGN;
  switch (d->mode) {
    case LKGD_A_PLAIN:
      if (d->csplit % 8 || d->csplit <= 0) return LKGD_E_SHAPE;
      if (d->csplit < d->K && !d->a1) return LKGD_E_NULL;
      if (d->csplit < d->K && d->csplit % BK) return LKGD_E_SHAPE;
      break;
    case LKGD_A_CONV3X3:
      if (d->Cin <= 0 || d->Cin % BK || d->K != 9 * d->Cin) return LKGD_E_SHAPE;
      if (d->csplit <= 0 || d->csplit % BK) return LKGD_E_SHAPE;
      if (d->csplit < d->Cin && !d->a1) return LKGD_E_NULL;
      if (d->Hout <= 0 || d->Wout <= 0 || d->Hin <= 0 || d->Win <= 0) return LKGD_E_SHAPE;
      if (d->stride != 1 && d->stride != 2) return LKGD_E_SHAPE;
      if (d->ups != 0 && d->ups != 1) return LKGD_E_SHAPE;
      if (d->M % (d->Hout * d->Wout)) return LKGD_E_SHAPE;
      // pad-1 output size must match: Hout = floor((Hv + 2 - 3)/stride) + 1
      if (d->Hout != (((d->Hin << d->ups) - 1) / d->stride) + 1) return LKGD_E_SHAPE;
      if (d->Wout != (((d->Win << d->ups) - 1) / d->stride) + 1) return LKGD_E_SHAPE;
      break;
    case LKGD_A_TCONV3:
      if (d->Cin <= 0 || d->Cin % BK || d->K != 3 * d->Cin) return LKGD_E_SHAPE;
      if (d->F <= 0 || d->HW <= 0 || d->Floc <= 0 || d->f_off < 0 || d->f_off + d->Floc > d->F) return LKGD_E_SHAPE;
      if (d->M % (d->Floc * d->HW)) return LKGD_E_SHAPE;
      break;
    case LKGD_A_CONV3X3_C8:
      if (d->Cin != 8 || d->K != 128 || d->lda0 != 8 || d->stride != 1 || d->ups != 0) return LKGD_E_SHAPE;
      if (d->Hout != d->Hin || d->Wout != d->Win || d->M % (d->Hout * d->Wout)) return LKGD_E_SHAPE;
      break;
    default:
      return LKGD_E_MODE;
  }
  return LKGD_OK;
}

extern "C" int lkgd_gemm_f16(const lkgd_gemm_desc* d, lkgd_stream_t stream) {
  int rc = check_desc(d);
  if (rc != LKGD_OK) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lkgd_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS) !=
        hipSuccess)
      return LKGD_E_LAUNCH;
    attr_set = true;
  }
  int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN;
  long long nwg = (long long)tiles_m * tiles_n;
  if (nwg > 0x7fffffffLL) return LKGD_E_SHAPE;
  hipLaunchKernelGGL(lkgd_gemm_kernel, dim3((unsigned)nwg), dim3(256), GEMM_LDS, (hipStream_t)stream, *d, tiles_m,
                     tiles_n);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
