// LKGD latent-knowledge fuse (include/lkgd_hip.h section 18): the block the reference recomputes in every UNet forward
// (models/unet_spatio_temporal_condition.py:536-595) - grouped 1x1 convolutions of the CLIP embedding and of the domain / flow
// logits (interpolated 1000 -> 1024), a quaternion linear on their concatenation, a 256-point real DFT of each, quaternion
// linears on magnitudes and phases, a 257-bin inverse DFT (512 real samples), and the two-layer `fuse_sf` MLP whose output
// REPLACES the CLIP embedding.  ~1.3 M multiply-adds on [B, 1024] vectors, input-invariant over the Euler steps: it runs once
// per clip, one workgroup per batch entry, fp32 throughout (the reference's FFT has no half path either), everything between
// the input vectors and the output row in LDS.  Latency, not a roofline: ~35 us, of which ~25 are the 2-MB weight read of the
// first quaternion linear by ONE workgroup.
#include "common.h"

struct lk_params {
  const float *e, *d, *f;            // [B, 1024], [Bd, 1000], [Bd, 1000] (Bd = 1: broadcast, reference :544-546)
  int B, Bd;
  const float *wl, *wd, *wf;         // Conv1d(1024 -> 256, k = 1, groups = 256): [256][4]
  const float* ctx;                  // [256]
  const float *w_fuse, *b_fuse;      // Hamilton matrix [1024][512] (in, out), [512]
  const float *cmag, *cpha;          // learned spectrum context [129]
  const float *w_mag, *b_mag, *w_pha, *b_pha;   // [512][256] (in, out), [256]
  const float *l0m, *l0p;            // Linear(4 -> 1) on the last bin: 4 weights + bias
  const float *sf0_w, *sf0_b, *sf2_w, *sf2_b;   // [1024][256] (in, out), [256]; [256][1024] (in, out), [1024]
  half_t* out;                       // [B, ldo]
  int ldo;
};

#define LK_NT 256
#define LK_PI 3.14159265358979323846f

// y[o] = b[o] + sum_i x[i] * W[i][o], W row-major (in, out): consecutive threads read consecutive columns
template <int PER>
__device__ __forceinline__ void lk_matvec(const float* __restrict__ W, const float* __restrict__ b, const float* x, int nin,
                                          int nout, float* y, int t) {
  float acc[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) acc[u] = 0.f;
  for (int i = 0; i < nin; ++i) {
    const float xi = x[i];
#pragma unroll
    for (int u = 0; u < PER; ++u) acc[u] = fmaf(xi, W[(long long)i * nout + t + u * LK_NT], acc[u]);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) y[t + u * LK_NT] = acc[u] + b[t + u * LK_NT];
}

__global__ __launch_bounds__(LK_NT) void lk_fuse_kernel(const lk_params p) {
  __shared__ float s_in[3][1024];        // e, interp(d), interp(f)
  __shared__ float s_cat[1024];          // low | low_d | low_f | ctx, later spatial | freq
  __shared__ float s_tw[512][2];         // cos, sin of 2 pi j / 512
  __shared__ float s_sp[3][129][2];      // spectra (re, im)
  __shared__ float s_mp[2][512];         // magnitudes | phases of bins 0..127 (4 x 128 each)
  __shared__ float s_mag[256], s_pha[256];
  __shared__ float s_spec[257][2];
  __shared__ float s_h[256];
  __shared__ float s_last[2][4];         // bin 128: magnitudes, phases
  const int t = threadIdx.x, b = blockIdx.x;
  const int bd = p.Bd == 1 ? 0 : b;
  for (int i = t; i < 1024; i += LK_NT) {
    s_in[0][i] = p.e[(long long)b * 1024 + i];
    // F.interpolate(size = 1024, mode = "linear", align_corners = False) of a 1000-sample row (reference :537,:540)
    float src = ((float)i + 0.5f) * (1000.0f / 1024.0f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src, i1 = i0 + 1 < 1000 ? i0 + 1 : 999;
    const float w1 = src - (float)i0, w0 = 1.0f - w1;
    s_in[1][i] = w0 * p.d[(long long)bd * 1000 + i0] + w1 * p.d[(long long)bd * 1000 + i1];
    s_in[2][i] = w0 * p.f[(long long)bd * 1000 + i0] + w1 * p.f[(long long)bd * 1000 + i1];
  }
  for (int j = t; j < 512; j += LK_NT) {
    float sn, cs;
    sincospif((float)j * (1.0f / 256.0f), &sn, &cs);      // 2 pi j / 512
    s_tw[j][0] = cs; s_tw[j][1] = sn;
  }
  __syncthreads();
  {   // grouped 1x1 convolutions: output channel c = 4-tap weighted sum of inputs 4c .. 4c+3 (thread = channel)
    const float* ws[3] = {p.wl, p.wd, p.wf};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += s_in[s][4 * t + k] * ws[s][4 * t + k];
      s_cat[s * 256 + t] = a;
    }
    s_cat[768 + t] = p.ctx[t];
  }
  __syncthreads();
  // 256-point real DFT of low / low_d / low_f, bins 0..128: X[k] = sum_n x[n] (cos - i sin)(2 pi k n / 256)
  for (int q = t; q < 3 * 129; q += LK_NT) {
    const int s = q / 129, k = q - s * 129;
    float re = 0.f, im = 0.f;
    for (int n = 0; n < 256; ++n) {
      const int j = ((k * n) & 255) * 2;             // table of 512 entries: angle 2 pi (kn mod 256) / 256
      const float x = s_cat[s * 256 + n];
      re = fmaf(x, s_tw[j][0], re);
      im = fmaf(-x, s_tw[j][1], im);
    }
    s_sp[s][k][0] = re; s_sp[s][k][1] = im;
  }
  __syncthreads();
  // spatial = quaternion_linear(cat) (the Hamilton matrix is expanded on the host): 1024 -> 512
  float spatial[2];
  {
    float acc[2] = {0.f, 0.f};
    for (int i = 0; i < 1024; ++i) {
      const float xi = s_cat[i];
      acc[0] = fmaf(xi, p.w_fuse[(long long)i * 512 + t], acc[0]);
      acc[1] = fmaf(xi, p.w_fuse[(long long)i * 512 + t + LK_NT], acc[1]);
    }
    spatial[0] = acc[0] + p.b_fuse[t]; spatial[1] = acc[1] + p.b_fuse[t + LK_NT];
  }
  // magnitudes / phases: bins 0..127 -> [4 x 128], bin 128 apart (reference :555-577).  Sign convention of the two REAL bins
  // (DC, Nyquist): the direct DFT above accumulates sin(0) / sin(pi k) terms that are exactly +0, so im = +0 and a negative real
  // part has phase +pi - what torch.angle(torch.fft.rfft(x)) returns on the host as well (tests/test_oracle_golden.py pins it;
  // an FFT library that produced -0 there would give -pi, 2 pi w away through pha0 / the pha quaternion linear)
  for (int q = t; q < 4 * 129; q += LK_NT) {
    const int s = q / 129, k = q - s * 129;
    float m, ph;
    if (s < 3) {
      const float re = s_sp[s][k][0], im = s_sp[s][k][1];
      m = hypotf(re, im);
      ph = atan2f(im, re);
    } else {
      m = p.cmag[k]; ph = p.cpha[k];
    }
    if (k < 128) { s_mp[0][s * 128 + k] = m; s_mp[1][s * 128 + k] = ph; }
    else { s_last[0][s] = m; s_last[1][s] = ph; }
  }
  __syncthreads();
  lk_matvec<1>(p.w_mag, p.b_mag, s_mp[0], 512, 256, s_mag, t);
  lk_matvec<1>(p.w_pha, p.b_pha, s_mp[1], 512, 256, s_pha, t);
  __syncthreads();
  {
    float sn, cs;
    sincosf(s_pha[t], &sn, &cs);
    s_spec[t][0] = s_mag[t] * cs; s_spec[t][1] = s_mag[t] * sn;
    if (t == 0) {
      float m0 = p.l0m[4], p0 = p.l0p[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { m0 = fmaf(s_last[0][s], p.l0m[s], m0); p0 = fmaf(s_last[1][s], p.l0p[s], p0); }
      sincosf(p0, &sn, &cs);
      s_spec[256][0] = m0 * cs; s_spec[256][1] = m0 * sn;
    }
  }
  __syncthreads();
  // inverse real DFT of 257 bins -> 512 samples (the imaginary parts of bins 0 and 256 do not enter, as in irfft)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int n = t + u * LK_NT;
    float a = s_spec[0][0] + ((n & 1) ? -s_spec[256][0] : s_spec[256][0]);
    float acc = 0.f;
    for (int k = 1; k < 256; ++k) {
      const int j = (k * n) & 511;
      acc = fmaf(s_spec[k][0], s_tw[j][0], acc);
      acc = fmaf(-s_spec[k][1], s_tw[j][1], acc);
    }
    s_cat[512 + n] = (a + 2.0f * acc) * (1.0f / 512.0f);
  }
  s_cat[t] = spatial[0]; s_cat[t + LK_NT] = spatial[1];
  __syncthreads();
  // fuse_sf: Linear(1024 -> 256), LeakyReLU(0.1), Linear(256 -> 1024)
  {
    float acc = 0.f;
    for (int i = 0; i < 1024; ++i) acc = fmaf(s_cat[i], p.sf0_w[(long long)i * 256 + t], acc);
    acc += p.sf0_b[t];
    s_h[t] = acc > 0.f ? acc : 0.1f * acc;
  }
  __syncthreads();
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 256; ++i) {
      const float hi = s_h[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = fmaf(hi, p.sf2_w[(long long)i * 1024 + t + u * LK_NT], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) p.out[(long long)b * p.ldo + t + u * LK_NT] = (half_t)(acc[u] + p.sf2_b[t + u * LK_NT]);
  }
}

extern "C" int lkgd_lk_fuse(const float* e, const float* d, const float* f, int32_t B, int32_t Bd, const float* const* w,
                            void* out, int32_t ldo, lkgd_stream_t stream) {
  if (!e || !d || !f || !w || !out) return LKGD_E_NULL;
  if (B <= 0 || B > 65535 || (Bd != 1 && Bd != B) || ldo < 1024) return LKGD_E_SHAPE;
  for (int i = 0; i < 18; ++i)
    if (!w[i]) return LKGD_E_NULL;
  lk_params p;
  p.e = e; p.d = d; p.f = f; p.B = B; p.Bd = Bd;
  p.wl = w[0]; p.wd = w[1]; p.wf = w[2]; p.ctx = w[3]; p.w_fuse = w[4]; p.b_fuse = w[5]; p.cmag = w[6]; p.cpha = w[7];
  p.w_mag = w[8]; p.b_mag = w[9]; p.w_pha = w[10]; p.b_pha = w[11]; p.l0m = w[12]; p.l0p = w[13];
  p.sf0_w = w[14]; p.sf0_b = w[15]; p.sf2_w = w[16]; p.sf2_b = w[17];
  p.out = (half_t*)out; p.ldo = ldo;
  hipLaunchKernelGGL(lk_fuse_kernel, dim3((unsigned)B), dim3(LK_NT), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
