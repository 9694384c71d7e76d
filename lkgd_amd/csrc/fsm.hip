// Track-guided feature fuse of the FSM hook (reference patch/patch_FSM.py:380-441), HBM-bound row kernels.
//
// The reference gathers tokens of one batch entry at tracked points, zeroes invisible points, scatter_adds them onto
// the partner entry's grid and divides by the visible count (+1e-6).  Here the tracks are inverted once on the host
// into a CSR list per target cell (points in their original order), so one wave owns one output row: it sums that
// cell's contributors in fp32 in a fixed order (no atomics, no canvas memset, deterministic), normalises, adds the
// residual row and the folded cross-attention bias and writes fp16.  The same kernel without a CSR list is the
// de-interleave copy (even entries -> contiguous conv input) and the "hidden + src_fused" combine.
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 4;

__global__ __launch_bounds__(256) void fsm_rows_kernel(lkgd_fsm_desc d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * kRowsPerBlock + wave;
  if (row >= (long long)d.pairs * d.HW) return;
  const int pair = (int)(row / d.HW), cell = (int)(row - (long long)pair * d.HW);
  const half_t* a = (const half_t*)d.a;
  const half_t* res = d.res ? (const half_t*)d.res + ((long long)pair * d.r_pair_rows + d.r_off + cell) * d.ldr : nullptr;
  const half_t* bias =
      d.bias ? (const half_t*)d.bias + (long long)((pair * d.bias_mul + d.bias_add) / d.bias_div) * d.ldb : nullptr;
  half_t* out = (half_t*)d.out + ((long long)pair * d.o_pair_rows + d.o_off + cell) * d.ldo;
  int p0 = 0, p1 = 0;
  float den = 1.0f;
  if (d.csr_off) {
    p0 = d.csr_off[row];
    p1 = d.csr_off[row + 1];
    float cnt = 0.f;
    for (int i = p0; i < p1; ++i) cnt += d.vis[d.csr_pt[i]];       // wave-uniform, same order as the reference
    den = cnt + 1e-6f;
  }
  for (int c = lane * 8; c < d.C; c += 512) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (d.csr_off) {
      for (int i = p0; i < p1; ++i) {
        const int pt = d.csr_pt[i];
        if (d.vis[pt] == 0.f) continue;
        const half_t* src = a + ((long long)pair * d.a_pair_rows + d.a_off + d.gather_idx[pt]) * d.lda + c;
        const half8_t v = *(const half8_t*)src;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
      // the reference divides in the activation dtype and hands an fp16 tensor on: round once here
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)(half_t)(acc[j] / den);
    } else {
      const half8_t v = *(const half8_t*)(a + ((long long)pair * d.a_pair_rows + d.a_off + cell) * d.lda + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)v[j];
    }
    if (res) {
      const half8_t r = *(const half8_t*)(res + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)r[j];
    }
    if (bias) {
      const half8_t b = *(const half8_t*)(bias + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)b[j];
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *(half8_t*)(out + c) = o;
  }
}

}  // namespace

extern "C" int lkgd_fsm_rows(const lkgd_fsm_desc* d, lkgd_stream_t stream) {
  if (!d || !d->a || !d->out) return LKGD_E_NULL;
  if (d->pairs <= 0 || d->HW <= 0 || d->C <= 0 || (d->C % 8)) return LKGD_E_SHAPE;
  if (d->lda < d->C || d->ldo < d->C || (d->lda % 8) || (d->ldo % 8)) return LKGD_E_SHAPE;
  if (d->a_pair_rows < 0 || d->a_off < 0 || d->o_pair_rows < 0 || d->o_off < 0) return LKGD_E_SHAPE;
  if (!aligned16(d->a) || !aligned16(d->out)) return LKGD_E_ALIGN;
  if (d->res && (d->ldr < d->C || (d->ldr % 8) || !aligned16(d->res) || d->r_pair_rows < 0 || d->r_off < 0))
    return LKGD_E_SHAPE;
  if (d->bias && (d->ldb < d->C || (d->ldb % 8) || !aligned16(d->bias) || d->bias_div <= 0 || d->bias_mul < 0 ||
                  d->bias_add < 0))
    return LKGD_E_SHAPE;
  if (d->csr_off && (!d->csr_pt || !d->gather_idx || !d->vis || d->P <= 0)) return LKGD_E_NULL;
  const long long rows = (long long)d->pairs * d->HW;
  if (rows + 1 > 0x7fffffffLL) return LKGD_E_SHAPE;
  const unsigned grid = (unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  hipLaunchKernelGGL(fsm_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
  return hipGetLastError() == hipSuccess ? LKGD_OK : LKGD_E_LAUNCH;
}
