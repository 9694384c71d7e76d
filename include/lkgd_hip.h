/*
 * lkgd_hip.h - C-ABI of the MI355X (gfx950) kernels behind the SVD / LKGD denoising hot path.
 *
 * The reference (caoql98/LKGD) is pure Python: it has no FFI of its own.  Every FLOP of its hot path is an ATen
 * operator reached through diffusers modules, so the "interface each entry point replaces" is the ATen call made
 * at the cited reference line (paths relative to /root/reference; [EXT] = inside diffusers==0.27.2, un-vendored,
 * restated in oracle/blocks.py).  The Python binding a maintainer adds on the reference side is shown in
 * INTEGRATION.md (ctypes, raw device pointers + the current HIP stream).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless named h_*; activations are fp16, channels-last:
 *     a "token matrix" [T, C] with T = (batch*frame, y, x) row-major, C contiguous.
 *   - no allocation, no synchronisation, no global state inside; safe on any stream; graph-capturable (the debug
 *     knobs declared at the end of this file are the one exception and say so).
 *   - return value: 0 = launched; negative = LKGD_E_* (nothing was launched).
 */
#ifndef LKGD_HIP_H
#define LKGD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lkgd_stream_t; /* hipStream_t */

enum {
  LKGD_OK = 0,
  LKGD_E_NULL = -1,      /* required pointer is NULL */
  LKGD_E_SHAPE = -2,     /* shape violates the kernel's tiling contract */
  LKGD_E_ALIGN = -3,     /* pointer / leading dimension not 16-byte aligned */
  LKGD_E_MODE = -4,      /* unknown mode / flag */
  LKGD_E_LAUNCH = -5     /* hipLaunchKernel reported an error */
};

/* ---------------------------------------------------------------------------------------------------------------
 * 1. MFMA GEMM with implicit-convolution A operand and fused epilogue.
 *
 *    out[m, n] = s_acc * ( sum_k A(m,k) * W[n,k] + bias[n] + rowbias[idx(m), n] )
 *                + r1 * res1[m, n] + r2 * res2[m, n]                           (fp32 accumulate, fp16 store)
 *    with idx(m) = ((m / rb_d1) * rb_m1 + (m % rb_d2) + rb_c0) % rb_md
 *    GEGLU (geglu = h, h in {32, 80}): packed W rows interleave h hidden rows | their h gate rows (h = 80 selects the
 *                     256x320-tile kernel, h = 32 the 128-wide-tile kernels); out[m, j] = hidden * gelu_erf(gate),
 *                     out has N/2 columns.
 *
 *    A(m,k) by `mode`:
 *      LKGD_A_PLAIN     A = a0[m, k] for k < csplit else a1[m, k - csplit]        (Linear, 1x1 conv, concat input)
 *      LKGD_A_CONV3X3   k = ((ky*(Cin/64) + c/64)*3 + kx)*64 + c%64 (per kernel row and 64-channel chunk the three horizontal
 *                       taps are consecutive K-tiles); token m = (n, y, x) on the Hout x Wout grid reads source pixel
 *                       ((y*stride+ky-1) >> ups, (x*stride+kx-1) >> ups) of the Hin x Win grid, zero outside
 *                       (pad 1); ups=1 folds nearest-2x upsampling into the gather; channel c from a0 / a1 as above
 *      LKGD_A_TCONV3    k = kt*Cin + c; token m = (b, f, s) reads frame f+kt-1 (zero outside [0,F)); Conv3d (3,1,1);
 *                       m = (b*Floc + fl)*HW + s is output frame f = f_off + fl, source row (b*F + f+kt-1)*HW + s
 *      LKGD_A_CONV3X3_C8 conv3x3 with Cin == 8 (conv_in): one 16-byte chunk per tap, K padded to 128
 *
 *    Replaces (reference call sites):
 *      F.linear      - attention to_q/k/v/to_out, GEGLU proj + FF out, proj_in/out, time_emb_proj
 *                      [EXT attention.py / attention_processor.py], witnessed at patch/patch.py:440-445,543-569
 *      F.conv2d 3x3  - models/unet_spatio_temporal_condition_controlnet.py:431 (conv_in), :500 (conv_out),
 *                      ResnetBlock2D conv1/conv2, Downsample2D, Upsample2D [EXT resnet.py]
 *      F.conv2d 1x1  - ResnetBlock2D.conv_shortcut [EXT]
 *      F.conv3d      - TemporalResnetBlock conv1/conv2 (3,1,1) [EXT resnet.py]
 *      F.gelu, residual adds, AlphaBlender - fused epilogues [EXT attention.py GEGLU, resnet.py AlphaBlender]
 *    Contract: K % 64 == 0; for conv modes Cin % 64 == 0 and csplit % 64 == 0 (C8 mode: Cin == 8, K == 128);
 *              N % 4 == 0 (geglu = h: N % 4h == 0); all leading dimensions % 8 == 0; pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------------- */
enum { LKGD_A_PLAIN = 0, LKGD_A_CONV3X3 = 1, LKGD_A_TCONV3 = 2, LKGD_A_CONV3X3_C8 = 3 };

typedef struct lkgd_gemm_desc {
  const void* a0;      /* fp16 */
  const void* a1;      /* fp16 or NULL */
  const void* w;       /* fp16 [N][K] */
  const float* bias;   /* fp32 [N] or NULL */
  const void* rowbias; /* fp16 [*, ldrb] or NULL */
  const void* res1;    /* fp16 [M, ldr1] or NULL */
  const void* res2;    /* fp16 [M, ldr2] or NULL */
  void* out;           /* fp16 [M, ldc] */
  const void* zeros;   /* >= 16 bytes of device zeros (source of padded / out-of-image loads) */
  int32_t M, N, K;
  int32_t lda0, lda1, csplit;
  int32_t mode, Cin;
  int32_t Hout, Wout, Hin, Win, stride, ups;
  int32_t F, HW;           /* TCONV3: frames per batch entry in the SOURCE tensor, tokens per frame */
  int32_t Floc, f_off;     /* TCONV3 under frame sharding: output rows cover Floc frames per batch entry starting at
                              global frame f_off; the source a0 holds all F frames.  Unsharded: Floc = F, f_off = 0 */
  int32_t ldrb, rb_d1, rb_m1, rb_d2, rb_md, rb_c0;
  int32_t ldr1, ldr2, ldc;
  float s_acc, r1, r2;
  int32_t geglu;           /* 0 = off, 32 or 80 = GEGLU interleave width (see above) */
  int32_t pad_off;         /* CONV3X3: 0 = symmetric zero padding 1 (every UNet convolution); 1 = no padding at the top / left
                              and one zero row / column at the bottom / right - F.pad(x, (0,1,0,1)) + stride-2 conv with
                              padding 0, the VAE encoder's Downsample2D [EXT diffusers downsampling.py] */
  void* workspace;         /* optional fp32 scratch for split-K partial sums (NULL = never split).  Few-row problems */
  int64_t workspace_bytes; /* (M < 8192: the 18x32 / 9x16 levels, or a frame-sharded rank's slice) fill a fraction of
                              the CUs with 128x128 tiles; K is then cut into up to 16 slices whose fp32 partial tiles
                              go to [slice][M][N] here and a second kernel adds them in slice order and runs the
                              epilogue - deterministic, no atomics.  One GEMM at a time may use a workspace. */
  float* colstats;         /* optional: per-(row block, column PAIR) sums of the ROUNDED outputs, fp32 [ceil(M / blk)][N / 2][2]
                              = (sum, sum of squares) of columns 2c, 2c + 1 over the valid rows of block blk - the GroupNorm
                              statistics (groups are even-sized) of the tensor
                              this GEMM writes, taken while its rows pass through LDS (section 2: lkgd_groupnorm_stats_cols),
                              so the separate statistics pass over the tensor disappears.  blk = lkgd_gemm_colstats_block(d)
                              (256 or 32); must be NULL when that returns 0.  Deterministic (fixed-order sums, no atomics). */
  const float* ln_colsum;  /* optional fp32 [N]: LayerNorm of the A rows folded into this GEMM (SURVEY K8;
                              F.layer_norm in front of a Linear, patch/patch.py:416,600): a0 holds the UN-normalised rows, w
                              the weights with the LayerNorm's gamma folded in (bias its beta), ln_colsum[n] = sum_k w[n][k]; the
                              kernel takes every row's mean / rstd over its K values from the registers that hold the row and
                              applies  out = rstd * (acc - mean * ln_colsum[n]) + bias[n] ...  in the epilogue - the normalised
                              tensor is never written.  Plain A, K = the LayerNorm's width <= 320 (K % 64 == 0), no GEGLU:
                              the row-panel program; LKGD_E_SHAPE otherwise. */
  float ln_eps;            /* the LayerNorm's epsilon (with ln_colsum) */
  int32_t cs_rows;         /* rows per GroupNorm sample of the tensor's consumer when `colstats` is (or is about to be) asked for,
                              0 = unknown: the tile-form choice then only considers tile rows that divide it (a 192-row form
                              would otherwise void the column sums of a 512-row sample).  Set it before lkgd_gemm_colstats_block */
} lkgd_gemm_desc;

int lkgd_gemm_f16(const lkgd_gemm_desc* d, lkgd_stream_t stream);
/* rows per `colstats` block of the tile program lkgd_gemm_f16 will run for this descriptor (its colstats field is ignored);
 * 0 = that program does not produce column statistics */
int lkgd_gemm_colstats_block(const lkgd_gemm_desc* d);
/* tile columns of the widest tile program for an output of N columns: 320, or 256 where 256 divides N and 320-column tiles would
 * idle more than a tenth of their columns (N = 256, 512, 768: the temporal VAE decoder's widths, diffusers
 * AutoencoderKLTemporalDecoder block_out_channels (128, 256, 512, 512)) - introspection for tests and tools */
int lkgd_gemm_wide_tile_n(int N);

/* ---------------------------------------------------------------------------------------------------------------
 * 2. GroupNorm (32 groups) statistics + apply + SiLU on channels-last tokens.
 *    x = cat(x0[:, :c0], x1[:, :c1]) along channels (x1 may be NULL); group g = channels [g*C/32, (g+1)*C/32).
 *    Statistics span `rows_per_sample` consecutive tokens (spatial GN: H*W; temporal 5-D GN: F*H*W, i.e. ACROSS
 *    frames).  stats[sample][32][2] = (mean, rstd), fp32.  `partial` is scratch of nsamples*nchunks*32*2 floats
 *    (deterministic two-stage reduction, no atomics); nchunks is returned by lkgd_groupnorm_chunks().
 *    Replaces: F.group_norm + F.silu - ResnetBlock2D/TemporalResnetBlock norm1/norm2,
 *    TransformerSpatioTemporalModel.norm [EXT], conv_norm_out unet_..._controlnet.py:498-499.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_groupnorm_chunks(int64_t rows_per_sample, int32_t C);
int lkgd_groupnorm_stats(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                         int64_t nsamples, int64_t rows_per_sample, float eps, float* partial, float* stats,
                         lkgd_stream_t stream);
/* frame-sharded variant: raw per-(sample, group) fp32 (sum, sum of squares) of the LOCAL rows - all-reduced across the
 * frame group by the host (RCCL) - then finalised with the GLOBAL element count per group */
int lkgd_groupnorm_sums(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                        int64_t nsamples, int64_t rows_per_sample, float* partial, float* sums, lkgd_stream_t stream);
int lkgd_groupnorm_finalize(const float* sums, int64_t nsamples, double count_per_group, float eps, float* stats,
                            lkgd_stream_t stream);
/* ... from the raw sums of `nparts` ranks as ONE gathered buffer (the sums ride on the all-gather that carries the Conv3d halo
 * frames, lkgd_amd/dist.py): part r's sums of sample s at parts + r*part_stride + s*sample_stride floats; added in rank order in
 * fp64, so every rank of the frame group derives bitwise the same (mean, rstd) without an all-reduce */
int lkgd_groupnorm_finalize_parts(const float* parts, int32_t nparts, int64_t part_stride, int64_t nsamples,
                                  int64_t sample_stride, double count_per_group, float eps, float* stats, lkgd_stream_t stream);
/* statistics of x = cat(x0[:, :c0], x1[:, :c1]) from the column sums their producing GEMMs left (lkgd_gemm_desc.colstats):
 * cs0 / cs1 = [rows / blk][ldcs / 2][2] fp32 (column pairs) with blk0 / blk1 rows per block (rows_per_sample must be a
 * multiple of both; channel counts and group size even);
 * as_sums = 0 writes stats[sample][32][2] = (mean, rstd) like lkgd_groupnorm_stats, 1 writes the raw (sum, sum of squares)
 * like lkgd_groupnorm_sums (frame-sharded path).  Replaces the read pass of F.group_norm over the tensor. */
int lkgd_groupnorm_stats_cols(const float* cs0, int32_t blk0, int32_t ldcs0, int32_t c0, const float* cs1, int32_t blk1,
                              int32_t ldcs1, int32_t c1, int64_t nsamples, int64_t rows_per_sample, float eps,
                              int32_t as_sums, float* stats, lkgd_stream_t stream);
int lkgd_groupnorm_apply(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1,
                         int64_t nsamples, int64_t rows_per_sample, const float* stats, const float* gamma,
                         const float* beta, int32_t silu, void* out, int32_t ldo, lkgd_stream_t stream);
/* lkgd_groupnorm_apply over a table of row segments in ONE launch: segs = nseg x { const void* src; void* dst; int64 rows;
 * int64 sample; } in DEVICE memory - rows [0, rows) of the [rows, C] matrix at src (row stride ld, as dst) are normalised with
 * stats[sample] and written to dst; max_rows = the longest segment.  A frame-sharded rank normalises its own frames of every batch
 * entry and the raw boundary frames it received from its neighbours this way (the same arithmetic as lkgd_groupnorm_apply, bit for
 * bit). */
int lkgd_groupnorm_apply_segments(const void* segs, int32_t nseg, int64_t max_rows, int32_t C, int32_t ld, const float* stats,
                                  const float* gamma, const float* beta, int32_t silu, lkgd_stream_t stream);
/* the whole F.group_norm (+ F.silu) of a tensor in one call: out = silu?(groupnorm(x)).  Small maps - one (sample, group) =
 * rows x C/32 channels of fp16 within 48 KiB, the tensor within 10.5 MB, at least 64 (sample, group) pairs: the 9x16 level and
 * a sharded rank's 18x32 level - take ONE launch, a workgroup per pair
 * (read once, statistics in fp32 sums / fp64 variance as lkgd_groupnorm_stats, normalised out of LDS; equal to the three
 * launches up to the summation order of the statistics).  Otherwise lkgd_groupnorm_stats followed by lkgd_groupnorm_apply
 * (the same three launches, the same numbers).  `partial` as above, `stats` (nsamples*32*2 floats) receives (mean, rstd). */
int lkgd_groupnorm_silu(const void* x0, int32_t c0, int32_t ld0, const void* x1, int32_t c1, int32_t ld1, int64_t nsamples,
                        int64_t rows_per_sample, float eps, float* partial, float* stats, const float* gamma,
                        const float* beta, int32_t silu, void* out, int32_t ldo, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 3. LayerNorm over the channel dimension of a token matrix, optional row-indexed bias added BEFORE normalising
 *    (frame positional embedding: x + emb[idx(row)]), idx as in (1); C % 8 == 0, C <= 2048.  gamma == beta == NULL: no affine (the UNet folds
 *    gamma/beta of every LayerNorm into the Linear that consumes it: W' = W*diag(gamma), b' = b + W*beta).
 *    Replaces: F.layer_norm - BasicTransformerBlock.norm1/3, TemporalBasicTransformerBlock.norm_in/1/3
 *    (patch/patch.py:416,556,600,610,670) and `hidden_states_mix + emb` [EXT transformer_temporal.py].
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_layernorm(const void* x, int32_t ldx, int64_t T, int32_t C, const float* gamma, const float* beta,
                   float eps, const void* rowbias, int32_t ldrb, int32_t rb_d1, int32_t rb_m1, int32_t rb_d2,
                   int32_t rb_md, void* out, int32_t ldo, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 5c. The attention half of a temporal transformer block in one launch, C = 320 / 5 heads of 64 / F <= 16 (the 72x128 level):
 *        out = W_o . attention_over_frames( q, k, v ) + b_o + x + rowbias[idx(row)],   q | k | v = W_qkv . LN(x) + b_qkv
 *    x / out: [B*F*HW, 320] token rows, row = (b*F + f)*HW + pixel; every pixel attends over its own F frames (softmax scale
 *    1/8); LN without affine (folded into W_qkv / b_qkv when the stream is packed); rowbias (may be NULL): fp16 table, idx as
 *    in (1): ((row / rb_d1) * rb_m1 + row % rb_d2 + rb_c0) % rb_md - the block's one-token cross-attention folded to a table.
 *    wstream: the chunk stream lkgd_amd/packing.py::pack_tblock builds (W_qkv, b_qkv, W_o in the order the kernel's generated
 *    statement consumes them, 849 920 bytes).  Neither q | k | v nor the attention output is ever written.
 *    Replaces: TemporalBasicTransformerBlock `norm1 -> attn1 -> + hidden_states` (patch/patch.py:592-597 regroup, :610,
 *    :660-661; [EXT] diffusers Attention / F.scaled_dot_product_attention) and the `attn2` residual add of a one-token context.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_tattn_block_c320(const void* x, int32_t ldx, const void* wstream, const float* bo, const void* rowbias, int32_t ldrb,
                          int32_t rb_d1, int32_t rb_m1, int32_t rb_d2, int32_t rb_md, int32_t rb_c0, void* out, int32_t ldo,
                          int32_t B, int32_t F, int32_t HW, float eps, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 3c. LayerNorm + a 320 -> 960 projection in one launch (the 72x128 level):  out[T, 960] = W . LN(x) + b,  LN without affine
 *    (folded into W / b when the stream is packed), fp32 accumulation, fp16 in / out; the normalised rows are never written.
 *    wstream: the chunk stream lkgd_amd/packing.py::pack_ln_proj builds (30 tiles of 32 output rows: bias fragment + 20 k-step
 *    fragments, 645 120 bytes).  ldo % 8 == 0, out 16-byte aligned.
 *    Replaces: BasicTransformerBlock `norm1` + `attn1.to_q / to_k / to_v` as one fused [3C, C] projection
 *    (patch/patch.py:416, :440-445; [EXT] diffusers Attention).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_ln_qkv_c320(const void* x, int32_t ldx, int64_t T, const void* wstream, float eps, void* out, int32_t ldo,
                     lkgd_stream_t stream);
/* the same at the 36x64 level: 640 -> 1920 (60 tiles of 32 output rows, each a 21-KiB chunk (bias fragment + k-steps 0..19) and a
 * 20-KiB chunk (k-steps 20..39): 2 519 040 bytes from pack_ln_proj); also norm1 -> to_q | to_k | to_v of the temporal blocks there */
int lkgd_ln_qkv_c640(const void* x, int32_t ldx, int64_t T, const void* wstream, float eps, void* out, int32_t ldo,
                     lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 3b. LayerNorm + GEGLU feed-forward + output projection + residual(s) in one launch, C = 320 / inner 1280 (the 72x128
 *    level).  x' = x + rowbias[idx(row)] (idx = (row / rb_d1) % rb_md; rowbias may be NULL);
 *        out = s_acc * ( W2 . ( hidden * gelu(gate) ) + b2 + x' ) + r2 * res2,   [hidden | gate] = W1 . LN(x') + b1
 *    LN without affine (gamma / beta folded into W1 / b1 when the stream is packed), exact-erf GELU, fp32 accumulation, fp16
 *    in / out.  wstream: the chunk stream lkgd_amd/packing.py::pack_ff_fused builds (W1, b1, W2 in the order the kernel's
 *    generated loop consumes them, 2 539 520 bytes).  The [T, 1280] intermediate is never written.
 *    Replaces: BasicTransformerBlock `norm3 -> ff` + residual (patch/patch.py:551-580) and TemporalBasicTransformerBlock
 *    `norm_in -> ff_in` + residual / `norm3 -> ff` + residual + AlphaBlender (patch/patch.py:599-608, :670-680; [EXT]
 *    diffusers FeedForward / GEGLU).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_ff_fused_c320(const void* x, int32_t ldx, int64_t T, const void* rowbias, int32_t ldrb, int32_t rb_d1, int32_t rb_md,
                       const void* wstream, const float* b2, float eps, float s_acc, const void* res2, int32_t ldr2, float r2,
                       void* out, int32_t ldo, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 4. Spatial self-attention, head_dim 64, flash-style (online softmax, scores never materialised).
 *    q/k/v: fp16 token matrices; head h occupies columns [h*64, h*64+64) of each; batch entry n owns rows
 *    [n*S, (n+1)*S).  kv_batch_map (int32[nbatch], may be NULL = identity) selects which batch entry's K/V rows a
 *    query batch attends to - the joint attention `attn1n` of patch/patch.py:466-482 is the same kernel with the
 *    partner permutation.  scale = 1/8.
 *    Replaces: F.scaled_dot_product_attention in AttnProcessor2_0 [EXT], BasicTransformerBlock.attn1
 *    (patch/patch.py:440-445, 503-508).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_attn_spatial(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                      void* out, int32_t ldo, int32_t nbatch, int32_t S, int32_t heads, const int32_t* kv_batch_map,
                      float scale, lkgd_stream_t stream);
/* the same with Sq query rows and S key rows per batch entry (q / out hold nbatch * Sq rows, k / v nbatch * S): a frame-sharded
 * DiT rank's local queries against the keys / values gathered from all ranks (lkgd_amd/cogvideox.py) */
int lkgd_attn_spatial_qk(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                         int32_t ldo, int32_t nbatch, int32_t Sq, int32_t S, int32_t heads, const int32_t* kv_batch_map,
                         float scale, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 5. Temporal self-attention: for every (batch b, pixel s, head h) attend over the F frames (F <= 32, head_dim 64).
 *    q/out hold Fq query frames per batch entry, k/v hold Fk >= 1 key frames (Fq == Fk unless the frames of a clip
 *    are sharded over GPUs: local queries against the all-gathered keys/values).
 *    Rows of q/k/v/out are tokens (b, f, s) - the [B*F,S,C] <-> [B*S,F,C] regroup of patch/patch.py:592-597,
 *    682-684 is an index map here, never a copy.  kv_b_map as in (4) (temporal joint branch :616-658).
 *    Replaces: TemporalBasicTransformerBlock.attn1 SDPA [EXT].
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_attn_temporal(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                       void* out, int32_t ldo, int32_t B, int32_t Fq, int32_t Fk, int32_t S, int32_t heads,
                       const int32_t* kv_b_map, float scale, lkgd_stream_t stream);

/* 5b. Fused temporal-attention front: out = attn1(norm1(x)) without its out-projection - LayerNorm (no affine: folded into
 *    the weights), the Q|K|V projection and the attention over the F frames of every (pixel, head) in one kernel; the
 *    normalised tokens and the Q|K|V rows never reach HBM.  C = 320 channels = heads x 64, F <= 16, HW % 16 == 0.
 *    wpack = the fused [3C, C] projection (rows q | k | v, head-major) re-laid as MFMA fragments:
 *    [head][q,k,v][4 fragments of 16 rows][10 K-steps of 32][lane = 16*(k/8 % 4) + row % 16][8]  (lkgd_amd/packing.py::
 *    pack_tfront); bqkv = fp32 [3C] or NULL.  Replaces, per temporal block: F.layer_norm + three F.linear + SDPA and the
 *    [B*F,S,C] <-> [B*S,F,C] regroups (patch/patch.py:592-597, :610, :660-661). */
int lkgd_tattn_front(const void* x, int32_t ldx, const void* wpack, const float* bqkv, void* out, int32_t ldo, int32_t B,
                     int32_t F, int32_t HW, int32_t heads, float eps, float scale, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 6. Loop glue (pipeline_stable_video_diffusion_trans.py:549-553, :578-592; scheduler :264-288, :418-528).
 *    lkgd_prepare_unet_input: latents fp16/fp32 [B,F,4,H,W] -> CFG duplicate, divide by sqrt(sigma^2+1), concat
 *      image_latents [2B|B,F,4,H,W] on channels, emit channels-last fp16 tokens [cfg*B*F*H*W, 8].
 *    lkgd_cfg_euler_step: noise_pred tokens [cfg*B*F*H*W, 4] (channels-last fp16) -> per-frame CFG
 *      uncond + g[f]*(cond-uncond), v-prediction x0, Euler update in fp32, write latents [B,F,4,H,W] (same dtype
 *      as input latents).  sigma / sigma_next are host scalars (the scheduler tables live on the host).
 *    lkgd_tokens_to_nchw: channels-last tokens [N*H*W, C] -> [N, C, H, W] fp16 (UNet.forward return layout).
 *    lkgd_nchw_to_tokens: the inverse for UNet.forward's `sample` argument ([N, C, H, W] -> tokens, any C<=ldo).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_prepare_unet_input(const void* latents, int32_t latents_is_f32, const void* image_latents, int32_t B,
                            int32_t F, int32_t H, int32_t W, int32_t cfg, float sigma, void* tokens_out,
                            lkgd_stream_t stream);
int lkgd_cfg_euler_step(const void* noise_tokens, void* latents, int32_t latents_is_f32, const float* guidance,
                        int32_t B, int32_t F, int32_t H, int32_t W, int32_t cfg, float sigma, float sigma_next,
                        int32_t prediction_type /*0 eps, 1 v*/, lkgd_stream_t stream);
/* Frame-sharded ranks: rows of this rank's [fl, HW, C] fp16 slice regrouped by destination pixel shard (pack != 0: the send
 * buffer of the all-to-all that re-shards the temporal attention by pixels - row (f, p) goes to fl*p0[r] + f*px[r] + (p - p0[r]),
 * r = the shard owning pixel p, px[0..k-1] = pixels per shard, k <= 16, sum = HW) or back (pack == 0: dst is the [fl, HW, C]
 * slice).  One launch instead of k strided copies on either side of every exchange.  px is a HOST array (read before return).
 * Replaces: the chunk / cat re-layout around torch.distributed.all_to_all_single in sequence-parallel attention
 * (CogVideo-main/tools/parallel_inference/parallel_inference_xdit.py:47-52 is the reference's only statement of it). */
int lkgd_shard_rows(const void* src, void* dst, int32_t fl, int32_t HW, int32_t C, int32_t k, const int32_t* px, int32_t pack,
                    lkgd_stream_t stream);
int lkgd_tokens_to_nchw(const void* tokens, int32_t ld, int64_t N, int32_t C, int32_t HW, void* out,
                        lkgd_stream_t stream);
int lkgd_nchw_to_tokens(const void* nchw, int64_t N, int32_t C, int32_t HW, void* tokens, int32_t ld,
                        lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 7. Small elementwise helpers used by the embedding path (unet_..._controlnet.py:406-419):
 *    lkgd_timestep_embedding: out[i, :] = [cos(t_i * f_j), sin(t_i * f_j)], f_j = exp(-ln(1e4) j / (dim/2)), fp16 out
 *    lkgd_silu: y = x * sigmoid(x) elementwise fp16;  lkgd_add: y = a + b elementwise fp16.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_timestep_embedding(const float* t, int32_t n, int32_t dim, void* out, int32_t ldo, lkgd_stream_t stream);
int lkgd_silu(const void* x, void* y, int64_t n, lkgd_stream_t stream);
int lkgd_add(const void* a, const void* b, void* y, int64_t n, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 8. Stand-alone scheduler ops for callers that use EulerDiscreteScheduler.scale_model_input / .step directly
 *    (utils/scheduling_euler_discrete_karras_fix.py:264-288, :418-528) on tensors of any shape (n elements).
 *    lkgd_scale: y = fp16(x * s).  lkgd_euler_step: prev = fp16(Euler(model_output fp16, sample fp16|fp32)).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_scale(const void* x, void* y, int64_t n, float s, lkgd_stream_t stream);
int lkgd_euler_step(const void* model_output, const void* sample, int32_t sample_is_f32, void* prev, int64_t n,
                    float sigma, float sigma_next, int32_t prediction_type, lkgd_stream_t stream);
/* the stochastic ("churn") form of the same step, s_churn > 0 (scheduling_euler_discrete_karras_fix.py:485-497):
 * sample += fp16(fp16(noise * s_noise) * churn) with churn = sqrt(sigma_hat^2 - sigma^2), sigma_hat = sigma * (gamma + 1),
 * then the Euler step from sigma_hat to sigma_next.  noise: fp16, n elements, drawn by the caller (randn_tensor). */
int lkgd_euler_step_churn(const void* model_output, const void* sample, int32_t sample_is_f32, const void* noise, void* prev,
                          int64_t n, float sigma, float sigma_hat, float s_noise, float churn, float sigma_next,
                          int32_t prediction_type, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 9. FSM hook row kernel (patch/patch_FSM.py:380-441, the track-guided fuse between even "src" and odd "dst" batch
 *    entries).  Token rows are addressed per PAIR of entries:
 *        row_x(pair, cell) = pair * x_pair_rows + x_off + cell,   cell in [0, HW)
 *    out[row_o] = value + res[row_r] + bias[(pair * bias_mul + bias_add) / bias_div]           (res, bias optional)
 *    value = a[row_a(pair, cell)]                                             when csr_off == NULL  (copy / combine)
 *    value = fp16( sum_{i in [csr_off[pair*HW+cell], csr_off[pair*HW+cell+1])}  vis[p] != 0 ? a[row_a(pair, gather_idx[p])] : 0
 *                  / (sum_i vis[p] + 1e-6) ),  p = csr_pt[i]                    otherwise
 *    which is torch.gather + masked zero + scatter_add + count-normalise of :405-418 / :429-437 with the tracks inverted
 *    on the host: csr_pt lists, per target cell, the global point ids (pair * P + k) in increasing order, so the fp32
 *    sum runs in the order of a sequential scatter_add; gather_idx / vis are indexed by global point id.
 *    All feature pointers fp16 with leading dimensions in elements (multiples of 8), C % 8 == 0, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct lkgd_fsm_desc {
  const void* a;
  const void* res;
  const void* bias;
  void* out;
  const int32_t* csr_off;    /* [pairs*HW + 1] or NULL */
  const int32_t* csr_pt;     /* [pairs*P] */
  const int32_t* gather_idx; /* [pairs*P] source cell of each point */
  const float* vis;          /* [pairs*P] visibility (0 = masked, value is the count weight) */
  int64_t a_pair_rows, a_off, r_pair_rows, r_off, o_pair_rows, o_off;
  int32_t lda, ldr, ldb, ldo;
  int32_t bias_mul, bias_add, bias_div;
  int32_t pairs, HW, C, P;
} lkgd_fsm_desc;
int lkgd_fsm_rows(const lkgd_fsm_desc* desc, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 10. Direct 3x3 convolution (pad 1, stride 1 or 2) for small channel counts, channels-last fp16, optional SiLU:
 *     the conditioning-embedding stack of the ControlNet-SVD encoder (models/controlnet_sdv.py:64-119).
 *     in [nimg*Hin*Win, ldi] (Cin % 8 == 0 channels used), w [Cout][3][3][Cin] fp16, bias fp32 [Cout] or NULL,
 *     out [nimg*Hout*Wout, ldo], Hout = (Hin-1)/stride + 1; Cout % 16 == 0.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_conv3x3_small(const void* in, int32_t Cin, int32_t ldi, const void* w, const float* bias, void* out,
                       int32_t Cout, int32_t ldo, int64_t nimg, int32_t Hin, int32_t Win, int32_t stride,
                       int32_t silu, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 11. Clip-level image pre-processing of the CLIP branch (boundary stage, once per clip):
 *     `_resize_with_antialiasing` of pipeline/pipeline_stable_video_diffusion_trans.py:661-765 =
 *       separable Gaussian blur with reflect padding (`_gaussian_blur2d` :752-765, `_filter2d` :713-733: one 1-D pass
 *       along W, one along H; the taps are `_gaussian` :736-749, computed by the caller)
 *       + `F.interpolate(mode="bicubic", align_corners=True)` (:686).
 *     fp32 planes [planes][H][W] (planes = batch * channels).  `lkgd_conv1d_reflect`: out[p,y,x] = sum_j taps[j] *
 *     in[p, reflect(y or x + j - (ntaps-1)/2)], axis 1 = along W, 0 = along H; in != out.
 *     `lkgd_resize_bicubic_ac`: cubic convolution A = -0.75, source index dst*(in-1)/(out-1), border clamp.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_conv1d_reflect(const float* in, float* out, int64_t planes, int32_t H, int32_t W, const float* taps,
                        int32_t ntaps, int32_t axis, lkgd_stream_t stream);
int lkgd_resize_bicubic_ac(const float* in, int64_t planes, int32_t H, int32_t W, float* out, int32_t Ho, int32_t Wo,
                           lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 12. Clip-level VAE stages (boundary stages either side of the loop): the two ops of
 *     AutoencoderKLTemporalDecoder [EXT diffusers 0.27.2; called at pipeline_stable_video_diffusion_trans.py:205-226,
 *     :256-283] that the GEMM / GroupNorm entry points above do not cover.
 *     `lkgd_softmax_rows`: y[r, :cols] = softmax(x[r, :cols]) per row, fp16 in / out, fp32 inside; cols % 8 == 0,
 *       cols <= 16384.  Replaces the softmax inside F.scaled_dot_product_attention of the single-head (head_dim 512)
 *       mid-block attention, whose QK^T and PV products run through lkgd_gemm_f16.
 *     `lkgd_time_conv_out`: out[b,f,co,p] = bias[co] + sum_{kt,ci} w[co][ci][kt] * x[(b, f+kt-1, p), ci], 3 channels,
 *       zero beyond the chunk's frames; input channels-last tokens [nbatch*F*HW, ld >= 4], output NCHW frames
 *       [nbatch*F, 3, HW] fp32 or fp16.  Replaces TemporalDecoder.time_conv_out (Conv3d (3,1,1)) + the layout change.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_softmax_rows(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t cols, lkgd_stream_t stream);
int lkgd_time_conv_out(const void* tokens, int32_t ld, const float* w, const float* bias, void* out, int32_t out_is_f32,
                       int64_t nbatch, int32_t F, int32_t HW, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 13. Front end of the domain / flow ViT encoders (train_models/train_svd_lora.py:1455-1466):
 *     `F.interpolate(images, size=[S,S], mode="bilinear")` (align_corners=False) fused with the P x P patch unfold of timm's
 *     PatchEmbed convolution [EXT]: out[(n, py, px), c*P*P + ky*P + kx] = resized[n, c, py*P+ky, px*P+kx], fp16
 *     [nimg*(S/P)^2, C*P*P] = the A operand of the patch-embedding GEMM (weight [D, C, P, P] flattened).  in: fp32 NCHW.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_vit_patchify(const float* in, int64_t nimg, int32_t C, int32_t H, int32_t W, void* out, int32_t S, int32_t P,
                      lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 14. DiT glue (the CogVideoX blocks; CogVideo-main/finetune/models/cogvideox_i2v/
 *     cogvideox_transformer_3d.py:126-158): everything else of a block is lkgd_gemm_f16 / lkgd_layernorm (rows up to 2048
 *     channels) / lkgd_attn_spatial.
 *     `lkgd_gelu_tanh`: F.gelu(approximate="tanh") of the feed-forward [EXT diffusers GELU], fp16, n % 8 == 0, in place allowed.
 *     `lkgd_gated_add`: out = res + gate[g(row)] * x, gate fp32 [2 * batch, C]: the adaLN-zero gates of the text stream (rows
 *       [0, split) of every batch entry's rows_per_batch rows) and of the video stream (the rest) - `hidden_states + gate_msa *
 *       attn_hidden_states` / `encoder_hidden_states + enc_gate_msa * ...` (:139-140,:155-156) in one pass over the joint buffer.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_gelu_tanh(const void* x, void* y, int64_t n, lkgd_stream_t stream);
int lkgd_gated_add(const void* x, int32_t ldx, const float* gate, const void* res, int32_t ldr, void* out, int32_t ldo,
                   int64_t rows, int32_t C, int32_t rows_per_batch, int32_t split, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 15. Cross-attention against a multi-token context, head_dim 64: the literal attn2 of BasicTransformerBlock
 *     (patch/patch.py:526-549) and TemporalBasicTransformerBlock (patch/patch.py:660-668) - F.scaled_dot_product_attention of
 *     AttnProcessor2_0 [EXT] with Lk > 1 keys.  (For SVD's one-token context the UNet folds attn2 into a row bias and never
 *     calls this.)  q / out: [T, heads*64] token rows; k / v: [ncontexts*Lk, heads*64], context c owns rows [c*Lk, (c+1)*Lk);
 *     row m attends to context idx(m) = ((m / rb_d1) * rb_m1 + m % rb_d2 + rb_c0) % rb_md, the row map of (1).
 *     ncontexts * Lk <= 256.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_attn_cross(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out, int32_t ldo,
                    int64_t T, int32_t heads, int32_t ncontexts, int32_t Lk, int32_t rb_d1, int32_t rb_m1, int32_t rb_d2,
                    int32_t rb_md, int32_t rb_c0, float scale, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 16. Dense self-attention for short sequences, any head_dim <= 128 (multiple of 8): the attention of the CLIP-ViT-H image
 *     encoder (`self.image_encoder(image).image_embeds`, pipeline/pipeline_stable_video_diffusion_trans.py:164-203 - 257 tokens,
 *     16 heads of 80 channels [EXT transformers CLIPAttention]).  q, k, v, out: [nbatch*S, heads*head_dim] token rows (row
 *     strides in halfs), softmax(q k^T * scale) v per (batch entry, head), fp32 arithmetic.  K and V of one head must fit the
 *     CU's LDS: 2 * S * (head_dim + 2) * 2 + 16 * S + 1024 <= 163840 bytes, else LKGD_E_SHAPE (longer sequences at head_dim 64:
 *     lkgd_attn_spatial).
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_attn_dense(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out, int32_t ldo,
                    int32_t nbatch, int32_t S, int32_t heads, int32_t head_dim, float scale, lkgd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 17. LKGD latent-knowledge fuse (models/unet_spatio_temporal_condition.py:536-595, parameters :197-225):
 *     e [B, 1024] CLIP embedding, d / f [Bd, 1000] domain / flow ViT logits (Bd = 1: broadcast over the batch, :544-546), all
 *     fp32 -> out [B, 1024] fp16, the embedding that REPLACES the CLIP embedding (:595, :613).  Inside: F.interpolate(size = 1024,
 *     "linear") of d / f, three Conv1d(1024 -> 256, k = 1, groups = 256), QuaternionLinear(1024 -> 512) on their concatenation with
 *     the learned context, rfft (256 -> 129 bins) of each, abs / angle, QuaternionLinear(512 -> 256) on the magnitudes and on the
 *     phases of bins 0..127, Linear(4 -> 1) x 2 on bin 128, irfft of the 257 bins (512 samples), fuse_sf = Linear(1024 -> 256) +
 *     LeakyReLU(0.1) + Linear(256 -> 1024).  One workgroup per batch entry, fp32; step-invariant, so a caller runs it once per clip.
 *     w[18], all fp32, matrices (in, out) row-major (quaternion layers as their expanded Hamilton matrices):
 *       0-2 lconv / dconv / fconv [256][4]; 3 texts [256]; 4 fuse W [1024][512], 5 bias [512]; 6 texts_fft_mag [129], 7 texts_fft_pha
 *       [129]; 8 fuse_fft_mag W [512][256], 9 bias; 10 fuse_fft_pha W, 11 bias; 12 fuse_fft_mag0 (4 weights + bias), 13 fuse_fft_pha0;
 *       14 fuse_sf.0 W [1024][256], 15 bias; 16 fuse_sf.2 W [256][1024], 17 bias.
 * ------------------------------------------------------------------------------------------------------------- */
int lkgd_lk_fuse(const float* e, const float* d, const float* f, int32_t B, int32_t Bd, const float* const* w, void* out,
                 int32_t ldo, lkgd_stream_t stream);

/* version / build info: "lkgd_hip <n> gfx950" */
const char* lkgd_version(void);

/* The debug / measurement knobs (lkgd_debug_set_*: forced tile programs, chunk sizes - the A/B tools under tools/ and the
 * per-variant parity tests use them) are NOT part of this interface: they are declared in include/lkgd_hip_debug.h, which no
 * product caller includes, and their state is per host thread (round 6), so the promise above - no process-global state, safe
 * to call from several host threads - holds whether or not somebody turns a knob. */

#ifdef __cplusplus
}
#endif
#endif /* LKGD_HIP_H */
