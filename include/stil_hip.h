/* libstil_hip.so -- C ABI of the MI355X (gfx950) kernels behind the STiL training step.
 *
 * The reference (kgutjahr/STiL-TTA) is pure PyTorch: it has no FFI, its "plugin API" is
 * LightningModule.training_step.  Each entry point below therefore replaces the ATen op
 * sequence issued at the cited reference lines; stil_tta_amd/ops.py binds them 1:1 (ctypes)
 * and stil_tta_amd/stil_model.py re-assembles STiLModel.training_step from them.
 *
 * Conventions
 *   - all tensors fp32, contiguous unless a leading dimension (ld*) is given; device pointers;
 *   - caller owns every buffer (PyTorch allocator); the library allocates nothing and keeps
 *     no mutable global state apart from a one-time kernel attribute (tuning choices are per-call arguments);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t), never synchronises;
 *   - return 0 = ok, <0 = error (-1 invalid argument, -2 unsupported shape, -3 HIP error);
 *     stil_last_error() returns the thread-local message; nothing throws across the boundary;
 *   - `accumulate` != 0 means "+=" into the destination gradient slot;
 *   - reductions use fixed-order sums (no float atomics): results are run-to-run bit-stable.
 *
 * NHWC activation layout: a feature map [N,H,W,C] is the row-major matrix [N*H*W, C].
 */
#ifndef STIL_HIP_H
#define STIL_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* stil_last_error(void);
int stil_version(void);
int stil_device_count(void);

/* ---- GEMM / convolution (fp32-exact MFMA, v_mfma_f32_32x32x2_f32) -------------------------
 * C[M,N] = act( (alpha * Agather[M,K] . W[N,K]^T - sub[n]) * scale[n] + shift[n] + bias[n] + resid[m,n] )
 * Agather row m = (n, oy, ox) over an NHWC source [*, srcH, srcW, srcC] (row stride lda),
 * k = (ky*KW + kx)*srcC + c.  mode 0: iy = oy*stride - pad + ky (forward);  mode 1: iy = (oy + pad - ky)/stride
 * when divisible (input-gradient of a strided conv).  A plain GEMM is srcH=srcW=OH=OW=KH=KW=1, srcC=K.
 * Output row map: row m = (n, oy, ox) is written to row (n*out_OH + oy*out_stride + out_py)*out_OW + ox*out_stride + out_px
 * (out_stride 1 = identity): the input-gradient of a stride-s conv is s*s such launches, one per output phase, each a
 * stride-1 gather over its own tap subset (stil_conv_weight_layout_phase) -- no multiply-by-zero work.
 * `pre` (optional) receives the value before the activation.  act: 0 none, 1 ReLU, 2 GELU(erf).
 * `colstats` (optional, raw product only): [2*cdiv(M,T), N] per-tile column statistics of C over tiles of
 * T = stil_gemm_nt_tile_rows(M,N) rows -- row 2t = mean, row 2t+1 = sum of squared deviations over tile t's valid rows --
 * which stil_bn_train_fwd_tiles turns into the batch statistics without a second pass over C.
 * Replaces nn.Conv2d / nn.Linear forward and input-gradient: models/resnets.py:112-132,248-260,
 * models/Transformer.py:27-33,63-88, STiLModel_backbone.py:19-32,139,153-155. */
int stil_gemm_nt(const float* A, const float* W, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                 int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad_y, int pad_x,
                 int mode, int out_stride, int out_py, int out_px, int out_OH, int out_OW,
                 const float* bias, const float* sub, const float* scale, const float* shift, const float* resid,
                 int ldr, float* pre, int act, float alpha, float* colstats, const float* a_bn, const float* relu_mask, int ldm,
                 const float* bs_y, const float* bs_stats, float* bstats, int bs_relu, int bs_tile0,
                 const float* scale_var, float var_eps, void* split_ws, size_t split_ws_bytes, int tune, void* stream);
/* `split_ws` (optional): split-K for grids below one 64x64 workgroup per CU (small per-GPU batches: the step is then a dependent
 *   chain of GEMMs that each leave most of the chip idle).  stil_gemm_nt_split_workspace_bytes(M, N, K, tune) > 0 says the product
 *   would be split; given a workspace of at least that size (256-byte aligned, ZEROED ONCE before its first use -- its first 16 KB
 *   hold one arrival ticket per tile, which every launch leaves zero again; products of different shapes may share it -- and not
 *   shared by launches that may run concurrently: one per stream),
 *   the tile's slices write their accumulators to slabs and the slice that draws the last ticket adds them in slice order and runs
 *   the epilogue: deterministic, no workgroup waits.  Without a workspace the product runs unsplit. */
size_t stil_gemm_nt_split_workspace_bytes(int M, int N, int K, int tune);
/* measurement hook (tests/tools/split_sweep.py): s > 0 forces s slices wherever a product can be split at all, s <= 0 = the policy */
int stil_gemm_nt_force_splits(int s);   /* returns the previous setting */
/* `scale_var` (optional, needs `scale`): the per-column scale is scale[n] / sqrtf(scale_var[n] + var_eps) -- an eval-mode BatchNorm
 *   folded into the conv epilogue straight from (weight, bias, running_mean, running_var): sub = running_mean, scale = weight,
 *   scale_var = running_var, shift = bias, formed exactly as stil_bn_eval_affine forms it (models/resnets.py:112-132 in eval mode). */
/* `bstats` (optional; 64x64 tiles with the 16-byte epilogue: stil_gemm_nt_bstats_ok): this launch produces the gradient g w.r.t.
 *   the output z = [relu](BN(y)) of a training conv+BN layer (it is the input-gradient GEMM of that layer's consumer).  The
 *   epilogue then also leaves the BatchNorm-BACKWARD partial sums of that layer, per 64-row tile t = bs_tile0 + tile index:
 *   bstats[(2t)*N + c] = sum_rows g'[., c], bstats[(2t+1)*N + c] = sum_rows g'[., c] * xhat[., c], xhat = (y - mean) * rstd,
 *   g' = g (bs_relu 0: g already carries the ReLU mask through relu_mask, or the layer has no ReLU) or g * [(y - mean)*a + beta > 0]
 *   (bs_relu 2).  bs_y: that layer's raw conv output, shaped and indexed like C (row stride ldc); bs_stats: its statistics block
 *   [4][N].  stil_bn_train_bwd_tiles consumes them: the reduction pass of BatchNorm backward over g and y (models/resnets.py:
 *   112-132, nn.BatchNorm2d backward) disappears.  bs_tile0 lets the phase launches of a strided input-gradient fill one array. */
int stil_gemm_nt_bstats_ok(const float* C, int N, int ldc, const float* resid, int ldr, const float* relu_mask, int ldm, const float* bs_y);
/* `relu_mask` (optional, row stride ldm, shaped like C): the stored value is zeroed where relu_mask <= 0.  The input-gradient
 *   GEMM of a block's first conv writes the gradient of the PREVIOUS block's output already multiplied by that output's ReLU
 *   mask (models/resnets.py:129-130: out += identity; out = relu(out)), so that block's BatchNorm backward neither reads z
 *   nor materialises the masked gradient. */
/* `a_bn` (optional; forward gathers with srcC % 16 == 0, 16-byte aligned operands): the A source is the RAW output y of a
 *   training conv+BN+ReLU layer and a_bn that layer's statistics block [4][srcC] (mean, rstd, a = gamma * rstd, beta, as
 *   stil_bn_train_fwd_tiles writes it); z = relu((y - mean) * a + beta) is formed while A is staged, so the inner layers of a
 *   residual block never materialise z (models/resnets.py:112-132: bn1/relu, bn2/relu fused into conv2 / conv3).  Always
 *   64x64 tiles with 16-deep k-tiles: `tune` must leave the tile automatic or ask for 11 (colstats is sized by tile rows).
 * `tune` (0 = automatic; otherwise for A/B measurements) = variant + 100 * bk32 + 1000 * acc2:
 *   variant  block tile: 22 = 128x128, 21 = 128x64, 12 = 64x128, 11 = 64x64 (0: automatic = 64x64, the fastest on the shapes of the step);
 *   bk32     32-deep LDS k-tiles instead of 16: 0 = automatic (= 3 wherever the reduction allows: K % 32 == 0 and 1x1 or Cin % 32 == 0;
 *            never with a_bn), 1 = wherever possible in two LDS buffers, 2 = never, 3 = wherever possible in ONE LDS buffer (64x64 tiles);
 *   acc2     two-level accumulation (partial chains of 64 products added to a master accumulator, ~ATen-CPU's
 *            rounding noise for long reductions): 0 = for K >= 512, 1 = never, 2 = always.
 *   + 10000  scalar epilogue (one dword per lane) instead of the 16-byte one 64x64 tiles use when N, ldc, ldr % 4 == 0 and
 *            every output-side pointer is 16-byte aligned (accumulators transposed through LDS; identical results).
 *   + 100000 the OPT-IN split-precision mode ("bf16x3"): where a launch qualifies (64x64 tiles, 16-byte aligned operands, K % 32 == 0,
 *            whole 32-channel taps) every staged fp32 operand element is split into three bf16 terms and a 32-deep k-tile is
 *            2 x 6 v_mfma_f32_32x32x16_bf16 (fp32 accumulate, two-level sums) instead of 16 v_mfma_f32_32x32x2_f32; launches that
 *            do not qualify run the fp32-exact kernels.  Distance from a float64 product at or below the fp32 chain's
 *            (profiles/r05_bf16x3_lab.txt); results differ from the fp32-exact mode in the last bits.  Never the bench's `value`.
 * Tile order (speed only, identical bits): row-tile major, or -- wide short products whose W exceeds an XCD's L2 (N >= 1024,
 * K <= 1024, N K 4 B > 2 MB) -- column panels of <= 1 MB of W (STIL_GEMM_PANEL = -1 automatic | 0 off | n forced).
 * rows per output tile / tile variant stil_gemm_nt uses for an [M,N] output under `tune` (colstats granularity, bench bookkeeping) */
int stil_gemm_nt_tile_rows(int M, int N, int tune);
int stil_gemm_nt_variant(int M, int N, int tune);
/* the instantiation stil_gemm_nt launches for these operands, as a CONFIG code whose digits are NOT the tune digits:
 * variant + 100 * bkd (0 = 16-deep k-tiles, 1 = 32-deep in two LDS buffers, 2 = 32-deep in one) + 1000 * acc2 (1 = two-level
 * sums, 0 = single chain) + 10000 * vec (16-byte loads; unrelated to tune's "+ 10000 scalar epilogue") + 100000 * plain (`plain` = 1: 1x1 / stride 1 / no padding / identity output map -- A is a plain row-major matrix
 * and the kernel's geometry code is compiled out) + 1000000 * a_bn (the operand-staging BatchNorm instantiation) + 10000000 * b3 (the
 * split-precision instantiation, tune + 100000) */
int stil_gemm_nt_config(const float* A, const float* W, int M, int N, int K, int lda, int ldb, int srcC, int KH, int KW, int plain, int a_bn, int tune);

/* Weight gradient  dW (+)= dY[M,N]^T . Xgather[M,K]  (split over M, slab partials + ordered reduce).
 * KH*KW > 1: dW is written in the reference layout (N, srcC, KH, KW); else [N, Kdst] (first Kdst columns).
 * `tune`: 0 = automatic block tile, 22 = 128x128 (64x128 for N <= 64), 11 = 64x64 (same value for the workspace query). */
size_t stil_wgrad_workspace_bytes(int M, int N, int K, int tune);
int stil_wgrad_tn(const float* dY, const float* X, float* dW, int M, int N, int K, int ldy, int ldx,
                  int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad,
                  int Kdst, int accumulate, const float* x_bn, float* workspace, size_t workspace_bytes, int tune, void* stream);
/* `x_bn` (optional): as stil_gemm_nt's a_bn, for the X operand (the raw conv output of the producing layer and its
 * statistics block [4][srcC]). */
/* DEFERRED reductions (small per-GPU batches under hipGraph replay: the step is ~1000 dependent launches of ~5 us): the _partial
 * entry points leave only the slab partials ([stil_wgrad_splits][N][K] / [stil_colsum_chunks][N]) in a workspace the CALLER keeps
 * until it has passed one StilReduceJob per product to stil_reduce_jobs, which finishes all of them in ceil(n / 48) launches -- bit
 * for bit what stil_wgrad_tn / stil_colsum write.  A colsum is the job {splits = chunks, N = 1, K = N, Cin = N, taps = 1, Kdst = N}.
 * Jobs of one call must have distinct `dst`.  `jobs` is HOST memory (the jobs travel as kernel arguments). */
typedef struct StilReduceJob {
  const float* P; float* dst;
  int splits, N, K, Cin, taps, Kdst, accumulate;
  float scale;
} StilReduceJob;
int stil_reduce_job_bytes(void);
int stil_wgrad_splits(int M, int N, int K, int tune);
int stil_wgrad_force_splits(int s);   /* measurement hook (tests/tools/split_sweep.py): s > 0 forces s slabs; returns the previous setting */
int stil_wgrad_tn_partial(const float* dY, const float* X, int M, int N, int K, int ldy, int ldx,
                          int srcH, int srcW, int srcC, int OH, int OW, int KH, int KW, int stride, int pad,
                          const float* x_bn, float* workspace, size_t workspace_bytes, int tune, void* stream);
int stil_colsum_chunks(int M);
int stil_colsum_partial(const float* X, int M, int N, int ld, float* workspace, size_t workspace_bytes, void* stream);
int stil_reduce_jobs(const void* jobs, int njobs, void* stream);

/* out[n] (+)= scale * sum_m X[m,n]   (bias gradients) */
size_t stil_colsum_workspace_bytes(int M, int N);
int stil_colsum(const float* X, float* out, int M, int N, int ld, int accumulate, float scale,
                float* workspace, size_t workspace_bytes, void* stream);

/* (Cout,Cin,KH,KW) -> [Cout][KH*KW][Cin] (forward operand) and [Cin][KH*KW][Cout] (dgrad operand) */
int stil_conv_weight_layout(const float* w, float* w_fwd, float* w_dgrad, int Cout, int Cin, int KH, int KW,
                            void* stream);
/* dgrad phase operand: w_sub[ci][ky'][kx'][co] = w[co][ci][ky0 + s*(KHs-1-ky')][kx0 + s*(KWs-1-kx')] */
int stil_conv_weight_layout_phase(const float* w, float* w_sub, int Cout, int Cin, int KH, int KW, int stride,
                                  int ky0, int kx0, int KHs, int KWs, void* stream);
/* Every per-step weight re-layout of a model in ONE launch (the three maps above: kind 0 = w_fwd, kind 1 = w_dgrad /
 * transpose, kind 2 = dgrad phase).  jobs: device array of n records of stil_weight_layout_job_bytes() bytes
 *   { int64 src, dst (float offsets into src_base / dst_base); int32 kind, Cout, Cin, KH, KW, stride, ky0, kx0, KHs, KWs,
 *     first_block, pad }
 * blk2job[b] = job of grid block b (block b is the (b - first_block)-th of its job; a job has
 * stil_weight_layout_job_blocks(kind, ...) blocks: 32x32 transpose tiles for kind 1, (output channel, 128-channel chunk) pairs
 * for kind 0 (KH*KW <= 9), 1024 destination elements for kind 2).
 * Replaces the per-layer calls of stil_conv_weight_layout / _phase / stil_transpose on the step path
 * (models/resnets.py:112-132 nn.Conv2d weights; nn.Linear weights of models/Transformer.py, disentangle_transformer.py). */
int stil_weight_layouts(const float* src_base, float* dst_base, const void* jobs, const int* blk2job, int n_blocks, void* stream);
int stil_weight_layout_job_bytes(void);
int stil_weight_layout_job_blocks(int kind, int Cout, int Cin, int KH, int KW, int KHs, int KWs);
int stil_im2col_nchw(const float* x, float* col, int N, int Cin, int H, int W, int OH, int OW, int KH,
                     int KW, int stride, int pad, int Kp, void* stream);
int stil_transpose(const float* in, float* out, int R, int C, void* stream);

/* ---- BatchNorm2d (NHWC rows), ReLU, residual, max-pool: models/resnets.py:112-132,248-252 ----
 * stats: [4,C] = mean, rstd, a=gamma*rstd, beta (z = (x-mean)*a + beta); bn_eval_affine's ab: [3,C] = a, beta, running_mean.  Train forward also updates the running
 * statistics (momentum, unbiased variance) and num_batches_tracked. z = relu?(x*a + b + resid).
 * `resid_stats` (optional, stil_bn_train_fwd_tiles): `resid` is the RAW conv output of the block's shortcut and resid_stats its
 * statistics block [4][C]; the shortcut's BatchNorm is applied inside this pass (the downsample branch never materialises its
 * normalised output, models/resnets.py:126-129).
 * stil_bn_train_fwd_tiles with z == NULL computes the statistics only (running buffers, `stats`): the consumer applies
 * them while it stages its operand (stil_gemm_nt a_bn / stil_wgrad_tn x_bn).
 * bn_train_bwd's relu: 0 = none, 1 = mask (z > 0) read from z, 2 = mask recomputed from x and stats (only without
 * a residual input; z may then be NULL).  gout (optional) receives dz*mask, the gradient of the residual branch. */
size_t stil_bn_workspace_bytes(int M, int C);
int stil_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, long long* num_batches_tracked, const float* resid, float* z,
                      float* stats, int M, int C, int relu, float eps, float momentum, float* workspace,
                      size_t workspace_bytes, void* stream);
int stil_bn_train_fwd_tiles(const float* x, const float* tilestats, int tile_rows, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* num_batches_tracked,
                            const float* resid, const float* resid_stats, float* z, float* stats, int M, int C, int relu,
                            float eps, float momentum, void* workspace, size_t workspace_bytes, void* stream);
size_t stil_bn_tiles_workspace_bytes(int M, int C, int tile_rows);
int stil_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float* ab, int C, float eps, void* stream);
int stil_bn_train_bwd(const float* dz, const float* z, const float* x, const float* gamma,
                      const float* stats, float* dx, float* gout, float* dgamma, float* dbeta, float* coef,
                      int M, int C, int relu, int accumulate, float* workspace, size_t workspace_bytes,
                      void* stream);
/* BatchNorm backward whose reduction pass was done by the GEMM that produced dz (stil_gemm_nt `bstats`): tilestats
 * [2*ntiles][C] (row 2t = sum g', row 2t+1 = sum g' * xhat over tile t) are combined in double in a fixed order, then the dx
 * pass of stil_bn_train_bwd runs (relu: the mask the sums were formed with).  nn.BatchNorm2d backward, models/resnets.py:112-132. */
size_t stil_bn_bwd_tiles_workspace_bytes(int ntiles, int C);
int stil_bn_train_bwd_tiles(const float* dz, const float* z, const float* x, const float* gamma, const float* stats,
                            const float* tilestats, int ntiles, float* dx, float* dgamma, float* dbeta, float* coef,
                            int M, int C, int relu, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int stil_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, int OH,
                          int OW, void* stream);
int stil_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C,
                          int OH, int OW, void* stream);

/* ---- transformer pieces: models/Transformer.py:63-88,165-174,240-259; disentangle_transformer.py:49-169 */
int stil_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd,
                       int rows, int D, float eps, void* stream);
size_t stil_layernorm_bwd_workspace_bytes(int rows, int D);
int stil_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean_rstd,
                       float* dx, float* dgamma, float* dbeta, int rows, int D, int accumulate,
                       float* workspace, size_t workspace_bytes, void* stream);
/* qkv: [B,T,3,H,d]. Queries are tokens [q_off,q_off+Sq), keys/values tokens [kv_off,kv_off+Skv).
 * out: [B,T,H*d] (query rows only); probs: [B,H,Sq,Skv] softmax before dropout; mask: optional keep-mask.
 * Backward ACCUMULATES into dqkv (zero it before the first window). */
int stil_attention_fwd(const float* qkv, float* out, float* probs, const unsigned char* mask, int B, int T,
                       int H, int d, int q_off, int Sq, int kv_off, int Skv, float scale, float drop_p,
                       void* stream);
int stil_attention_bwd(const float* dout, const float* qkv, const float* probs, const unsigned char* mask,
                       float* dqkv, int B, int T, int H, int d, int q_off, int Sq, int kv_off, int Skv,
                       float scale, float drop_p, void* stream);
/* kind 1: dx = dy*(ref>0) (ref = ReLU output); kind 2: dx = dy*gelu'(ref) (ref = pre-activation) */
int stil_act_bwd(const float* dy, const float* ref, float* dx, long n, int kind, void* stream);
/* out = resid + x * emask[i] * rmask[i/rowlen] * scale   (nn.Dropout / drop_path / residual add) */
int stil_drop_add(const float* x, const float* resid, const unsigned char* emask, const unsigned char* rmask,
                  float* out, long n, int rowlen, float scale, void* stream);
int stil_axpby(const float* x, const float* y, float* out, long n, float a, float b, void* stream);
/* keep-mask = hash(seed, *step_dev, offset + i) >= p; the per-step counter is read from DEVICE memory (bumped by
 * stil_counter_inc once per step) so that a captured hipGraph draws fresh masks on every replay */
int stil_rng_mask(unsigned char* out, long n, unsigned long long seed, unsigned long long offset, float p,
                  const unsigned long long* step_dev, void* stream);
int stil_counter_inc(unsigned long long* counter_dev, void* stream);
int stil_tab_embed_fwd(const float* x, const int* cat_offsets, const float* cat_emb, const float* con_w,
                       const float* con_b, const float* cls, const float* colemb, float* h, int B,
                       int ncols, int ncat, int D, void* stream);
size_t stil_tab_embed_bwd_workspace_bytes(int ncols, int D);
int stil_tab_embed_bwd(const float* g, const float* x, const int* cat_offsets, const int* rowcol,
                       int n_emb_rows, float* d_emb, float* d_con_w, float* d_con_b, float* d_cls,
                       float* d_colemb, int B, int ncols, int ncat, int D, int accumulate, float* workspace,
                       size_t workspace_bytes, void* stream);
int stil_tokmean_fwd(const float* x, float* y, int B, int T, int D, void* stream);
int stil_tokmean_bwd(const float* g, float* dx, int B, int T, int D, void* stream);

/* ---- SAINT tabular encoder (config_dvm_STiL_SAINT): STiLModel_SAINT_backbone.py:159-184, SAINT/model_util.py:43-59,79-122
 * token j < ncat+1 of out [B,nfeats,d]: embeds[(j ? int(x[b,cat_cols[j-1]]) : 0) + cat_offsets[j]] + pos_enc[j];
 * tokens tok0+j: the per-continuous-column simple_MLP(1 -> hid -> d), all columns in ONE launch (the reference loops in
 * Python); param_ptrs / grad_ptrs are device arrays of ncon*4 pointers {w1[hid], b1[hid], W2[d,hid], b2[d]}. */
int stil_saint_embed_fwd(const float* x, const int* cat_cols, const int* cat_offsets, const float* embeds,
                         const float* pos_enc, float* out, int B, int ncols, int ncat, int nfeats, int d, void* stream);
int stil_saint_embed_bwd(const float* g, const float* x, const int* cat_cols, const int* cat_offsets,
                         const int* rowcol, int n_emb_rows, float* d_embeds, float* d_pos, int B, int ncols,
                         int ncat, int nfeats, int d, int accumulate, void* stream);
int stil_colmlp_fwd(const float* x, const int* con_cols, const float* const* param_ptrs, float* out, int B,
                    int ncols, int ncon, int nfeats, int tok0, int hid, int d, void* stream);
int stil_colmlp_bwd(const float* g, const float* x, const int* con_cols, const float* const* param_ptrs,
                    float* const* grad_ptrs, int B, int ncols, int ncon, int nfeats, int tok0, int hid, int d,
                    int accumulate, void* stream);
/* GEGLU: out[r, :H] = h[r, :H] * gelu(h[r, H:]);  row softmax (inter-sample attention over the batch) */
int stil_geglu_fwd(const float* h, float* out, long rows, int H, void* stream);
int stil_geglu_bwd(const float* g, const float* h, float* dh, long rows, int H, void* stream);
int stil_row_softmax_fwd(const float* z, float* p, int rows, int C, void* stream);
int stil_row_softmax_bwd(const float* g, const float* p, float* dz, int rows, int C, void* stream);

/* ---- loss tail: STiLModel.py:259-303,339,374-381; utils/clip_loss.py; utils/prototype_loss.py; club.py */
int stil_ce_hard(const float* logits, int ld, const long long* labels, float* row_loss, float* dlogits,
                 int ldd, int rows, int K, float inv_rows, void* stream);
int stil_ce_soft(const float* logits, int ld, const float* targets, int ldq, const float* row_w,
                 float* row_loss, float* dlogits, int ldd, int rows, int K, float inv_rows, void* stream);
int stil_reduce_sum(const float* x, int n, float scale, float* out, int accumulate, void* stream);
int stil_scale_dev(const float* x, const float* g_dev, float c, float* out, long n, void* stream);
int stil_l2norm_fwd(const float* x, float* y, float* norms, int rows, int D, void* stream);
int stil_l2norm_bwd(const float* g, const float* y, const float* norms, float* dx, int rows, int D,
                    void* stream);
/* CLIPLoss.forward on the logits Z = n0 n1^T / T (utils/clip_loss.py:34-38): lse = [2][B] row / column log-sum-exps, kept in
 * DOUBLE -- the gradient dZ sums to zero over the whole matrix and the projector-bias gradient downstream is the image of
 * that total through the embeddings' common component, so an lse error shared by all rows is amplified by the batch size
 * (csrc/loss.hip); dZ is formed in double and rounded once. */
int stil_clip_fwd(const float* Z, double* lse, float* terms, float* loss, int B, float lam0, float lam1,
                  void* stream);
int stil_clip_bwd(const float* Z, const double* lse, const float* g_dev, float* dZ, int B, float lam0,
                  float lam1, void* stream);
int stil_club_fwd(const float* mu, const float* y, const float* ybar, float* rows_tmp, float* out2, int R,
                  int D, void* stream);
int stil_club_bwd(const float* mu, const float* y, const float* ybar, const float* mubar,
                  const float* g_club_dev, const float* g_est_dev, float* dmu, float* dy, int R, int D,
                  void* stream);
/* flags: [Bu,4] = {case id 1..4 (case1, case2_i, case2_t, case3), mask1, confident-for-prototypes, 0};
 * w3: [3,Bu] row weights of the unlabelled CE terms (multimodal, imaging, tabular). */
int stil_cgpl_pgls(const float* zm, const float* zi, const float* zt, int ldz, const float* feat_u,
                   const float* prototypes, const unsigned char* mask_random, float* pseudo_label,
                   float* pseudo_orig, float* prediction, unsigned char* flags, int* hard, float* w3,
                   const float* pred_in, int Bu, int K, int Dp, float rate_pseudo, float T, float th, int use_pseudo,
                   void* stream);
/* distribution alignment (STiLModel.py:171-180): out = (probs / queue_mean) renormalised per row; feed as pred_in above */
int stil_da_apply(const float* probs, const float* queue_mean, float* out, int rows, int K, void* stream);
int stil_proto_loss(const float* feat, const float* prototypes, const int* hard, const unsigned char* conf,
                    float* row_loss, float* dfeat_unit, int rows, int K, int Dp, float T, void* stream);
/* class_sum_cnt: [K, Dp+1] (last column = counts); labelled rows (< B_l) are divided by repeat_ratio */
int stil_proto_accum(const float* feat_e, const int* hard, const unsigned char* conf, float* class_sum_cnt,
                     int B, int B_l, int K, int Dp, float repeat_ratio, void* stream);
int stil_proto_add(const float* class_sum_cnt, float* prototypes_sum, float* prototypes_count_sum, int K,
                   int Dp, void* stream);
int stil_proto_commit(float* prototypes, float* prototypes_sum, float* prototypes_count_sum,
                      int* bad_count_dev, int K, int Dp, void* stream);

/* CoMatch's pseudo-label-graph contrastive loss (models/MatchModel/CoMatch.py:104-117) on S = f_s0 . [f_s1 ; queue_s]^T / T
 * [rows, N] and the pseudo-label graph Q [rows, N] (comatch_model.py:287-297): row_loss[r] = -sum_j log(p_rj + 1e-7) w_rj over
 * the edges Q >= threshold, p = exp(S) / rowsum(exp(S)), w = Q / rowsum(Q over the edges); dS (optional) = d(mean row_loss)/dS
 * with inv_rows = 1/rows. */
int stil_contrast_graph(const float* S, int lds, const float* Q, int ldq, float threshold, float* row_loss, float* dS,
                        int ldd, int rows, int N, float inv_rows, void* stream);
/* SimMatch's label unfolding / aggregation (models/MatchModel/simmatch_model.py:289-302): teacher[r,j] = tpo[r,j] * probs[r,
 * labels[j]] renormalised per row (tpo [rows, N] = softmax(feat_ku . bank / tt), labels [N] = the bank's class ids);
 * pseudo[r,k] (optional) = probs[r,k] * c_smooth + (sum of tpo[r,j] with labels[j] == k) * (1 - c_smooth), or probs when
 * c_smooth >= 1. */
int stil_simmatch_unfold(const float* tpo, const float* probs, const long long* labels, float* teacher, float* pseudo,
                         int rows, int N, int K, float c_smooth, void* stream);
/* FreeMatch (models/MatchModel/FreeMatchFolder/freematch_model.py:132-168): self-adaptive threshold state update + masking from
 * probs [rows, K] = softmax of the teacher's weak-view logits: time_p [1], p_model [K], label_hist [K] are updated in place
 * (EMA with `momentum`), mask[r] = max prob >= time_p * p_model[argmax] / max(p_model), onehot [rows, K] / idx [rows] = the hard
 * pseudo-labels.  scratch: rows floats. */
int stil_freematch_update(const float* probs, int rows, int K, float* p_model, float* label_hist, float* time_p,
                          float momentum, float* mask, float* onehot, int* idx, float* scratch, void* stream);
/* FreeMatch fairness loss (FreeMatchFolder/freematch_utils.py:18-47) over the rows with mask != 0 of the student's strong-view
 * logits [rows, K]: loss [1] and dlogits [rows, K] = d loss / d logits (zero outside the mask; both zero when the mask is empty).
 * Scratch: P [rows, K] floats, pred [rows] ints, vec [4 K] floats. */
int stil_freematch_entropy(const float* logits, const float* mask, int rows, int K, const float* p_model,
                           const float* label_hist, float* loss, float* dlogits, float* P, int* pred, float* vec, void* stream);
/* logged partition ratios (STiLModel.py:307-311) from cgpl_pgls's flags [rows,ld] = (case id 1..4, mask1, ...):
 * out5 = {threshold1_ratio, case1_ratio, case2_i_ratio, case2_t_ratio, case3_ratio} */
int stil_flag_ratios(const unsigned char* flags, int ld, int rows, float* out5, void* stream);

/* hard pseudo-labels of the MMatch baseline (models/SemiMultimodal/MMatch.py:223-226): onehot[r] = e_argmax(probs[r]) (first
 * maximum), row_mask[r] = (max >= threshold) as 0/1 floats, idx[r] = the argmax */
int stil_onehot_argmax(const float* probs, int rows, int K, float threshold, float* onehot, float* row_mask, int* idx,
                       void* stream);

/* ---- flat-slab EMA teacher update and Adam: STiLModel.py:154-168, 563-570 */
int stil_ema_update(float* ema, const float* model, long n, double momentum, void* stream);
/* the reference's EMA applied to int64 buffers (CoTraining_SAINT.py:102-105: float32 arithmetic, truncating copy):
 * ema[i] = (int64)(fl(float(ema[i]) * m) + fl(float(1 - m) * float(model[i]))) */
int stil_ema_int_trunc(long long* ema, const long long* model, int n, double momentum, void* stream);
int stil_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   const int* chunk2tensor, int* steps, const unsigned char* active, int n_tensors, long n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   void* stream);

/* ---- evaluation metrics (torchmetrics==0.11.0 as used by STiLModel.py:122-145, 360-363, 458-463, 529-545) ----
 * hits_total: device int64[2] = {hits, samples}, accumulated (integer atomics: order-independent).
 * metric_topk: target class within the k best scores of its row (k = 1: argmax's first-maximum rule).
 * metric_binary: (prob > threshold) == (target == 1).
 * auroc: exact area under the ROC curve (thresholds=None); K == 1: binary on scores[:,0] with positives target == 1;
 *        K > 1: one-vs-rest per class and their unweighted mean (macro; a class without positives or negatives
 *        scores 0).  Integer rank statistics per class (segmented radix sort), no float accumulation. */
int stil_metric_topk(const float* scores, int ld, const long long* target, int N, int K, int k,
                     long long* hits_total, void* stream);
int stil_metric_binary(const float* probs, const long long* target, int N, float threshold, long long* hits_total,
                       void* stream);
size_t stil_auroc_workspace_bytes(int N, int K);
int stil_auroc(const float* scores, int ld, const long long* target, int N, int K, float* auc_per_class, float* macro,
               void* ws, size_t ws_bytes, void* stream);

/* ---- device input pipeline ---------------------------------------------------------------
 * Batch versions of what the reference's DataLoader workers compute per sample on the CPU.  Random draws are
 * arguments (drawn by the host wrapper stil_tta_amd/augment.py, or injected by tests).
 *
 * stil_tab_corrupt: ContrastiveImagingAndTabularDataset.corrupt (datasets/ContrastiveImagingAndTabularDataset.py:146-158):
 *   out[b,:] = clean[b,:]; out[b, idx[b,j]] = marginal[idx[b,j], pos[b,j]], j < k = int(n_cols * corruption_rate);
 *   marginal [n_cols, n_rows] = the training table transposed (:121-127); idx rows hold DISTINCT columns (random.sample).
 * stil_tab_corrupt_draw: those draws on the device (partial Fisher-Yates + uniform rows, counter-based hash RNG;
 *   `step` = optional device-side step counter so that hipGraph replays draw fresh values). */
int stil_tab_corrupt(const float* clean, const float* marginal, const int* idx, const int* pos, float* out,
                     int B, int n_cols, int n_rows, int k, void* stream);
int stil_tab_corrupt_draw(int* idx, int* pos, int B, int n_cols, int n_rows, int k, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* step, void* stream);
/* stil_aug_resize: crop box[b] = (top, left, h, w) -> bilinear resize to P x P (half-pixel centres, no antialias) ->
 *   optional horizontal flip -> optional colour (jitter[b] = brightness, contrast, saturation, gray flag; torchvision's
 *   float formulas in that order) -> * scale -> clip [0,1] -> float CHW [B,3,P,P].  Source: uint8 HWC [B,H,W,3] or float
 *   CHW [B,3,H,W] (exactly one non-null).  Replaces RandomResizedCrop / Resize / HorizontalFlip / ColorJitter / ToGray /
 *   convert_to_ts of utils/utils.py:33-59 and ContrastiveImagingAndTabularDataset.py:177-198 (GaussianBlur: stil_aug_blur).
 * stil_aug_gray_mean: per-sample mean grey level of the brightness-adjusted source image (what adjust_contrast blends with). */
int stil_aug_gray_mean(const unsigned char* src_u8, const float* src_f32, const float* jitter, float* gmean, int B, int H,
                       int W, float scale, void* stream);
/* stil_aug_blur: separable Gaussian blur of the whole source image (transforms.GaussianBlur(kernel_size, sigma), utils/utils.py:52):
 *   torchvision's kernel weights, "reflect" borders; sigma[b] <= 0 copies sample b; -> float CHW [B,3,H,W] scaled by `scale`
 *   (tmp: same size, the row-pass intermediate). */
int stil_aug_blur(const unsigned char* src_u8, const float* src_f32, const float* sigma, float* tmp, float* out, int B, int H,
                  int W, int ksize, float scale, void* stream);
/* stil_aug_rotate: the whole source image rotated by angle[b] degrees (counter-clockwise) about its centre, same size, bilinear,
 *   borders mirrored without repeating the edge pixel (A.Rotate / cv2.BORDER_REFLECT_101; utils/utils.py:82,112,170) -> float CHW.
 * stil_aug_hue: torchvision's float adjust_hue in place on float CHW images in [0,1] (hue[b] in [-0.5, 0.5], may be null), then
 *   the 3-channel grey image where gray[b] != 0 (may be null): the tail of the strong-view colour jitter, utils/utils.py:225-226. */
int stil_aug_rotate(const unsigned char* src_u8, const float* src_f32, const float* angle, float* out, int B, int H, int W,
                    float scale, void* stream);
int stil_aug_hue(float* img, const float* hue, const float* gray, int B, int H, int W, void* stream);
int stil_aug_resize(const unsigned char* src_u8, const float* src_f32, const int* box, const unsigned char* flip,
                    const float* jitter, const float* gmean, float* out, int B, int H, int W, int P, float scale,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
