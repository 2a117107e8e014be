/*
 * espnet_amd C ABI — MI355X (gfx950) kernels for the hybrid CTC/attention ASR hot path.
 *
 * The reference (kan-bayashi/espnet v0.9.5) has no native interface for this path: it is pure
 * Python over torch ops plus the third-party warp-ctc wheel.  The boundary below is therefore the
 * one SURVEY.md §8b defines; every entry point cites the reference computation it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 / int32 / int64 data owned by the caller; the
 *    library never allocates, frees or retains memory and keeps no mutable global state;
 *  - every call is asynchronous on `stream` (hipStream_t passed as void*), re-entrant, and
 *    performs no host synchronisation (so it can be captured into a hipGraph);
 *  - return 0 on success, <0 for a rejected argument (EAMD_EINVAL=-1, EAMD_EUNSUPPORTED=-2),
 *    >0 = hipError_t from the launch.  Numerical failure (e.g. infeasible CTC alignment) is
 *    reported in the data (+inf loss), never as an error code.
 */
#ifndef ESPNET_AMD_H_
#define ESPNET_AMD_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int eamd_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM family (MFMA).  Replaces every nn.Linear / torch.matmul / pointwise Conv1d / Conv2d on the
 * path: transformer/attention.py:33-36,55-57,91,96,113,195,199; positionwise_feed_forward.py:25-32;
 * subsampling.py:28-35; conformer/convolution.py:28-53; decoder.py:247; ctc.py:26,85.
 *
 *   acc[m,n] = sum_k opA(A)[m,k] * opB(B)[k,n]
 *   v = acc + bias[n];  v = epilogue(v, aux[m,n]);  v = alpha*v + R[m,n] + beta*C_old[m,n]
 *
 * transA=0: A is [M,K] row-major (lda); transA=1: A is stored [K,M] (lda = stride of k).
 * transB=0: B is stored [N,K] (nn.Linear weight, ldb = stride of n); transB=1: B is [K,N].
 * a_act/b_act apply an activation to the operand while it is staged (EAMD_ACT_*).
 * epilogue: 0 none, 1 relu, 2 swish, 3 multiply by (aux>0), 4 multiply by dswish(aux), 5 multiply by aux,
 *   6 (only with Hb + dropout, below): C receives the factor d Hb / d v = mask / (1 - p) * h_act'(v) instead of v - the
 *   FFN forward then hands its backward a ready factor (epilogue 5 there) instead of the pre-activation.
 * splitk>1: partial sums are atomically ADDED to C (caller pre-initialises C; epilogue must be 0,
 *           beta ignored, bias/R contributed by split 0).
 * precision: 0 = fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 products),
 *            1 = bf16 MFMA (fp32 accumulate); operands are either fp32 in memory and rounded while
 *                staged (in_dtype 0) or already bf16 in memory (in_dtype 1, the fast path: 16-byte
 *                staging, BK = 64, ds_read_b64_tr_b16 for the k-strided operand layouts).
 * colsum (transA=1 only): the column sums of A are added there by the same launch, i.e. the bias
 * gradient dB = sum_m dY[m,:] comes for free with dW = dY^T X.
 * Two-level batch (batch1 x batch2) with independent element strides for A/B/C(aux,R share C's).
 * Optional implicit-im2col view of A (`gather`) and strided row map for C (`cmap`) express the
 * Conv2dSubsampling convolutions and their input-gradient without materialising columns.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t enabled;
  int32_t C;          /* channels per tap (contiguous in memory, NHWC) */
  int32_t ntap;
  int32_t Ho, Wo;     /* logical row = (b*Ho + i)*Wo + j */
  int32_t Hin, Win;   /* source tensor is [B,Hin,Win,C] */
  int32_t sh, sw;     /* source pixel = (i*sh + dh[tap], j*sw + dw[tap]); out of range -> 0 */
  int32_t dh[9], dw[9];
} eamd_gather_t;

typedef struct {
  int32_t enabled;
  int32_t Ho, Wo;     /* logical row = (b*Ho + i)*Wo + j */
  int32_t Hc, Wc;     /* physical row = (b*Hc + i*sh + oh)*Wc + j*sw + ow */
  int32_t sh, oh, sw, ow;
} eamd_rowmap_t;

typedef struct {
  const float* A; const float* B; float* C;
  const float* bias; const float* aux; const float* R;
  float* colsum;       /* optional, transA only: colsum[m] += alpha * sum_k A[k,m] (bias gradient) */
  int32_t M, N, K;
  int32_t transA, transB;
  int64_t lda, ldb, ldc, ldaux, ldr;
  int32_t batch1, batch2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  float alpha, beta;
  int32_t a_act, b_act, epilogue;
  int32_t splitk;
  int32_t precision;
  int32_t tile;        /* 0 auto, 64 or 128 */
  eamd_gather_t gather;
  eamd_rowmap_t cmap;
  void* Cb;            /* optional bf16 copy of the result (same layout as C); C may then be NULL */
  int32_t in_dtype;    /* 0: A,B are fp32; 1: A,B are bf16 (precision must be 1) */
  int32_t aux_dtype;   /* 0: aux is fp32; 1: aux is bf16 */
  /* fused dropout (bf16-operand kernel, splitk == 1): keep(i) = hash(drop_step[0], drop_salt, i) >= p * 2^32 with
   * i = row * ldc + col, the same mask eamd_dropout draws for a contiguous [M, N] tensor.
   *   Hb == NULL: v <- keep ? v / (1 - p) : 0 on the epilogue value (after bias / activation / aux factor,
   *               before alpha, residual and beta), e.g. out = x + scale * dropout(W2 h + b2);
   *   Hb != NULL: second bf16 output Hb = dropout(h_act(v)) while C / Cb keep v itself (FFN: z and
   *               h = dropout(act(z)) from one launch, positionwise_feed_forward.py:27). */
  float drop_p;
  uint64_t drop_salt;
  const uint64_t* drop_step;
  void* Hb;
  int32_t h_act;
  /* operand-side dropout (fp32-MFMA kernel, in_dtype 0 / precision 0): while an operand is staged, element i of its
   * contiguous [rows, ld] matrix becomes keep(i) ? a_act(x) / (1 - p) : 0 with the mask eamd_dropout draws for
   * (drop_step, salt, i) - a consumer reads dropout(act(z)) or dropout(dY) without that tensor ever being written:
   *   W2 (drop(act(z))) in the FFN forward, dW2 += dY^T drop(act(z)), every block's incoming-gradient dropout in
   *   backward (positionwise_feed_forward.py:27, encoder_layer.py:101-144).  Needs an unbatched, un-gathered operand. */
  float a_drop_p, b_drop_p;
  uint64_t a_drop_salt, b_drop_salt;
  int32_t h_dtype;     /* dtype of the second output Hb: 0 bf16, 1 fp32 (fp32-MFMA kernel: z and h = dropout(act(z)) in fp32) */
  /* epilogue 7 (row statistics; splitk 1, tile given as 64 or 128, no batch / row map): the result v = A B + bias is NOT
   * stored (C, Cb may be NULL); every workgroup leaves, for each row m of its tile and its column tile j,
   *   part[(m * ceil(N / tile) + j) * 2 + {0, 1}] = (max_n v[m,n], sum_n exp(v[m,n] - max)) over the tile's columns,
   * and the values of two columns per row: zcol[m] = v[m, col[m]] (col may be NULL or col[m] < 0: none) and
   * zfix[m] = v[m, fix].  A log-softmax over N = thousands of columns then needs the fp32 logits neither written nor
   * read back: eamd_rnnt_node_stats_part combines the partials (transducer loss: lse, log p(blank), log p(label) per
   * lattice node; reference: transducer/loss.py:74-76 hands the materialised logits to warp-transducer).
   * epilogue 8 (softmax-gradient rows; same restrictions, C and / or Cb required): with the per-row coefficients
   * rowc[m] = (tot, gb, gl) the stored result is
   *   sc * (exp(v[m,n] + tot) - [n == fix] * gb - [n == col[m]] * gl),   sc = scale * (gscale ? gscale[0] : 1),
   * and 0 for rows with tot = -inf: d(-log P) / d logits of a transducer lattice node (tot = log occupancy - lse, gb / gl the
   * blank / label transition posteriors; eamd_rnnt_row_coef fills rowc) written straight from the recomputed logits. */
  struct {
    float* part; const int32_t* col; float* zcol; float* zfix; int32_t fix; int32_t reserved;
    const float* rowc; const float* gscale; float scale; int32_t reserved2;
  } stats;
} eamd_gemm_t;

int eamd_gemm(const eamd_gemm_t* p, void* stream);

/* Up to EAMD_GEMM_MULTI_MAX INDEPENDENT products (no result of one is an operand or overlaps a result of another) issued
 * together: the stride-parity classes of a strided convolution's input gradient (reference: the autograd of
 * transformer/subsampling.py:28-35's second Conv2d) are implicit products of 1 - 4 taps over the same rows, and one launch
 * whose workgroups are dealt to the problems in turn keeps tiles of all reduction lengths resident together.  Every
 * descriptor is validated as eamd_gemm validates it BEFORE anything is launched.  One launch when all are fp32-operand,
 * precision 0, gathered x W^T products on the 128 x 128 tile (splitk 1, batch 1, epilogue <= 5, no dropout, no second
 * output); any other combination runs as n eamd_gemm calls in order.  Results are those of the separate calls, bit for bit. */
#define EAMD_GEMM_MULTI_MAX 4
int eamd_gemm_multi(const eamd_gemm_t* descs, int n, void* stream);

/* Grouped launch of independent weight-gradient GEMMs (the dW_i += alpha dY_i^T X_i of one backward pass - reference:
 * the autograd of every nn.Linear on the path, e.g. transformer/attention.py:30-33 - each too small to fill the chip on
 * its own): ONE kernel whose workgroups look their problem up in a device-resident table.
 * eamd_gemm_group_plan validates n HOST descriptors (all of one in_dtype / precision; transA = transB = 1, batch 1, no
 * bias / residual / aux / epilogue / gather / row map / dropout; split-K accumulating with atomics, or splitk = 1 with
 * beta = 1; colsum allowed; operands meeting the 16-byte staging conditions of eamd_gemm's fast kernels), writes the
 * first workgroup of every problem into first[0..n] and returns the total workgroup count, or EAMD_EUNSUPPORTED / EAMD_EINVAL.
 * All descriptors of one launch name the same tile (64, or 128 for outputs of at least 128 x 128: p.tile); workgroup counts are
 * padded to multiples of 8 so that every XCD works on a contiguous run of one problem's tiles x K-slices.
 * The caller copies the descriptors and `first` to the device (stream-ordered) and calls eamd_gemm_group_launch. */
int eamd_gemm_group_plan(const eamd_gemm_t* descs, int n, int32_t* first);
int eamd_gemm_group_launch(const eamd_gemm_t* descs_dev, const int32_t* first_dev, int n, int total, int in_dtype, int tile,
                           void* stream);

/* Fused position-wise feed-forward block.  reference: transformer/positionwise_feed_forward.py:12-32 (w_2(dropout(act(w_1(x))))),
 * wired as conformer/encoder_layer.py:96-103,139-146 / transformer/encoder_layer.py (x + ff_scale * dropout(ff(norm(x)))).
 * One workgroup takes 32 rows through BOTH products; the [M, F] hidden units are written once for backward and never
 * read back by the second product.
 *   eamd_ffn_fwd:  out[M,D] = R + alpha * drop_out( drop_in(act(x W1^T + b1)) W2^T + b2 )
 *                  h[M,F] <- drop_in(act(z)),  f[M,F] <- mask_in / (1 - p_in) * act'(z)   (either may be NULL: inference)
 *   eamd_ffn_bwd:  h[M,F] <- dz = alpha * (x W2) (.) f   (x = gradient of the block output [M,D], f as left by forward),
 *                  out[M,D] <- dz W1
 * Dropout masks are eamd_dropout's (element index of the contiguous [M,F] / [M,D] tensor, salts salt_in / salt_out,
 * drop_step = the device step counter of eamd_rng_advance).  w1 [F,D], w2 [D,F] row-major (nn.Linear layout).
 * In BOTH dtypes w1 / w2 address the PACKED weight images of eamd_ffn_pack_f32 / eamd_ffn_pack_bf16 (MFMA fragment order:
 * every wave-instruction of the kernels reads 1 KB of consecutive bytes - fetched from the nn.Linear layout a fragment load
 * touches 16 rows x 64 bytes and the kernels sit at the vector memory path's line rate): fwd_first / fwd_second for
 * eamd_ffn_fwd, bwd_first / bwd_second for eamd_ffn_bwd.  The pack entry points make the four images (F * D elements each)
 * of one layer from its nn.Linear-layout weights (w1 [F, D], w2 [D, F]) in one launch; re-run whenever the weights change
 * (every optimizer step).
 * dtype 0 = fp32 operands on v_mfma_f32_16x16x4_f32 (F a multiple of 128).
 * dtype 1 = bf16 operands on v_mfma_f32_16x16x32_bf16: x, f, h and the images address bf16 (b1, b2, R, out stay fp32); F a
 *   multiple of 256.
 * Returns EAMD_EUNSUPPORTED for shapes the kernels are not built for (D != 256, F not a multiple of 128 / 256, unaligned
 * operands): the caller then runs the two eamd_gemm products. */
typedef struct {
  const float* x; const float* w1; const float* b1; const float* w2; const float* b2; const float* R;
  float* out; float* f; float* h;
  int32_t M, D, F, act;
  float alpha;
  float p_in; uint64_t salt_in; float p_out; uint64_t salt_out; const void* drop_step;
  int32_t dtype;
  /* hsplit = 2 (fp32 operands, F a multiple of 256; 0 / 1 = off): for few rows (M / 32 workgroups fill half the chip or less,
   * e.g. the decoder's 3232 target positions) every 32-row block is served by TWO workgroups, each over half of the hidden
   * units; both ADD their share of the second product to `out`, which the caller has ZEROED (two addends onto zeros: the
   * result does not depend on their order).  h / f / dz are written per half as before. */
  int32_t hsplit;
  /* Optional LayerNorm in front, eamd_ffn_fwd only (the norm_ff / norm_ff_macaron of conformer/encoder_layer.py:96-103,139-146
   * and transformer/decoder_layer.py: x + s * dropout(ff(norm(x)))).  ln_x != NULL: the block input is ln_x [M, D] fp32 and
   * the workgroup normalises its 32 rows while it stages them (nn.LayerNorm arithmetic: biased variance, rsqrt(var + eps));
   * `x` is then an OUTPUT - the normalised rows in the operand dtype, which backward's weight gradient reads - together with
   * ln_mean / ln_rstd [M].  ln_w, ln_b [D] fp32.  ln_x == NULL: the other ln_* fields are ignored. */
  const float* ln_x; const float* ln_w; const float* ln_b; float* ln_mean; float* ln_rstd;
  float ln_eps; int32_t reserved2;
  /* Optional LayerNorm BACKWARD behind eamd_ffn_bwd (fp32 operands, no hsplit): lnb_x != NULL (the block input [M, D] of the
   * forward's LayerNorm, with lnb_gamma [D], lnb_mean / lnb_rstd [M]): `out` receives dx = LayerNorm'(dz W1) + lnb_dres instead
   * of dz W1, lnb_drop_out (optional, fp32 [M, D]) its dropped copy (probability lnb_drop_p, salt lnb_drop_salt, drop_step),
   * lnb_ws [ceil(M / 32)][2][D] the per-workgroup partial sums of d gamma / d beta (eamd_layernorm_bwd_reduce, nblk = ceil(M / 32));
   * same arithmetic as eamd_rowproj's lnb_* fields. */
  const float* lnb_x; const float* lnb_gamma; const float* lnb_mean; const float* lnb_rstd; const float* lnb_dres;
  float* lnb_ws; float* lnb_drop_out; uint64_t lnb_drop_salt; float lnb_drop_p; int32_t reserved3;
} eamd_ffn_t;
int eamd_ffn_fwd(const eamd_ffn_t* p, void* stream);
int eamd_ffn_bwd(const eamd_ffn_t* p, void* stream);
/* eamd_ffn_pack_f32 for MANY feed-forward blocks in one launch (fp32 operands, D = 256): a 12-layer macaron Conformer re-packs
 * 24 blocks every optimizer step.  Each job names one block's nn.Linear-layout weights and its four image buffers. */
typedef struct {
  const float* w1; const float* w2;
  float* fwd_first; float* fwd_second; float* bwd_first; float* bwd_second;
  int32_t D, F;
} eamd_ffn_pack_t;
int eamd_ffn_pack_f32_multi(const eamd_ffn_pack_t* jobs, int njobs, void* stream);

/* Row-block projection: out[M, N] = R + alpha * drop( A'[M, K] B[K, N] + bias ) with 32 rows per workgroup taken through the whole
 * product (csrc/rowproj_f32.hip; fp32 operands, exact fp32 products; K a multiple of 256 up to 768, N a multiple of 256).
 * Replaces, for the K = 256 / N = 256 products of a Conformer / Transformer block, the nn.Linear / pointwise Conv1d calls of
 * transformer/attention.py:40-61,90-92 (linear_q/k/v as one [3D, D] product, linear_out) and conformer/convolution.py:63,76
 * (pointwise_conv1, pointwise_conv2) TOGETHER WITH what surrounds them in conformer/encoder_layer.py:106-138:
 *   the LayerNorm in front          ln_x != NULL (K = 256): A' = LayerNorm(ln_x) formed while the rows are staged; `a` is then an
 *                                   OUTPUT (the normalised rows [M, 256], which backward's weight gradient reads), with ln_mean /
 *                                   ln_rstd [M] (transformer/layer_norm.py:12-38);
 *   BatchNorm apply + activation    a_scale != NULL (K = 256): A' = act(a * a_scale[k] + a_shift[k]) (convolution.py:73-75: the
 *                                   caller folds mean / rstd / gamma / beta into scale and shift); a_out [M, 256] (optional)
 *                                   receives A';
 *   dropout + residual behind       p_out / salt_out / drop_step (eamd_dropout's mask of the contiguous [M, N] result), alpha, R;
 *   the LayerNorm BACKWARD behind an input gradient   lnb_x != NULL (N = 256; the product is dxn = dy W): `out` receives
 *                                   dx = LayerNorm'(dxn) + lnb_dres, lnb_drop_out (optional, fp32 [M, 256]) its dropped copy
 *                                   (mask of salt lnb_drop_salt, probability lnb_drop_p: the incoming-gradient dropout of the
 *                                   PREVIOUS block), and lnb_ws [ceil(M / 32)][2][256] the per-workgroup partial sums of d gamma /
 *                                   d beta (second stage: eamd_layernorm_bwd_reduce with nblk = ceil(M / 32)).
 * `w` addresses the PACKED image of B made by eamd_rowproj_pack_f32 (MFMA fragment order; re-packed whenever the weights change):
 *   trans = 0: B[k][n] = W[n * ldw + k]  (y = x W^T, W [N, K] as nn.Linear stores it)
 *   trans = 1: B[k][n] = W[k * ldw + n]  (dx = dy W, W [K rows, N columns])
 * Any number of images is packed by one launch (a table of jobs travels in the kernel arguments, 48 per launch).
 * EAMD_EUNSUPPORTED for other shapes / unaligned operands: the caller runs eamd_gemm (+ eamd_layernorm_*). */
typedef struct {
  const float* a; int64_t lda;
  const float* w; const float* bias;
  const float* R; int64_t ldr;
  float* out; int64_t ldo;
  int32_t M, K, N, a_act;
  float alpha; float p_out; uint64_t salt_out; const void* drop_step;
  const float* ln_x; const float* ln_w; const float* ln_b; float* ln_mean; float* ln_rstd; float ln_eps; float lnb_drop_p;
  const float* a_scale; const float* a_shift; float* a_out;
  const float* lnb_x; const float* lnb_gamma; const float* lnb_mean; const float* lnb_rstd; const float* lnb_dres;
  float* lnb_ws; float* lnb_drop_out; uint64_t lnb_drop_salt;
} eamd_rowproj_t;
typedef struct {
  const float* w; float* image;      /* image: K * N floats */
  int32_t K, N, ldw, trans;
} eamd_rowproj_pack_t;
int eamd_rowproj(const eamd_rowproj_t* p, void* stream);
int eamd_rowproj_pack_f32(const eamd_rowproj_pack_t* jobs, int njobs, void* stream);
int64_t eamd_rowproj_lnb_workspace(int M);     /* floats of lnb_ws */
int eamd_ffn_pack_f32(const float* w1, const float* w2, float* fwd_first, float* fwd_second, float* bwd_first, float* bwd_second,
                      int D, int F, void* stream);
int eamd_ffn_pack_bf16(const void* w1, const void* w2, void* fwd_first, void* fwd_second, void* bwd_first, void* bwd_second,
                       int D, int F, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row kernels (HBM-bound).
 * ------------------------------------------------------------------------------------------ */
/* LayerNorm over the last dim.  reference: transformer/layer_norm.py:12-38 (eps = 1e-12). */
/* y (fp32) and/or y_bf16 (bf16 copy for the bf16-operand GEMM) are written; either may be NULL. */
int eamd_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean,
                       float* rstd, int rows, int D, float eps, void* stream);
/* dgamma/dbeta are ACCUMULATED (+=); dx = LN-backward(dy) + dres (dres optional, may alias dx: the
 * residual branch's gradient is folded in).  workspace: eamd_layernorm_bwd_workspace(rows, D) floats of
 * scratch for the two-stage column reduction (NULL => one f32 atomic per column per block instead). */
int64_t eamd_layernorm_bwd_workspace(int rows, int D);
/* Deferred second stage: with dgamma == dbeta == NULL (workspace required) eamd_layernorm_bwd / _bwd_drop leave the
 * per-block partial sums in `workspace` = [nblk][2D] floats, nblk = eamd_layernorm_bwd_workspace(rows, D) / (2D);
 * eamd_layernorm_bwd_reduce then ACCUMULATES the partials of any number of such passes into their dgamma / dbeta
 * with one launch per 64 jobs (the table is read on the host during the call; the workspaces must stay alive until
 * the launch has run).  The reference leaves this sum to autograd (nn.LayerNorm backward, layer_norm.py:12-38). */
typedef struct {
  const float* ws; float* dgamma; float* dbeta;
  int32_t nblk, D;
} eamd_ln_reduce_job_t;
int eamd_layernorm_bwd_reduce(const eamd_ln_reduce_job_t* jobs, int njobs, void* stream);
int eamd_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                       float* workspace, int rows, int D, void* stream);
/* Same, with a second output dx_drop_bf16 = bf16(dropout(dx; p, salt)) drawn with the mask eamd_dropout uses for a
 * contiguous [rows, D] tensor: the incoming-gradient dropout + cast of the block BEFORE this LayerNorm (whose output
 * dropout the forward applied with (p, salt)), fused into this kernel's store.  D = 256 or 512, 16-byte aligned
 * operands; otherwise EAMD_EUNSUPPORTED (run eamd_layernorm_bwd and eamd_dropout instead). */
int eamd_layernorm_bwd_drop(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                            const float* dres, float* dx, void* dx_drop_bf16, float drop_p, const uint64_t* step_dev,
                            uint64_t salt, float* dgamma, float* dbeta, float* workspace, int rows, int D, void* stream);
/* fp32 twin: the dropped copy is fp32 (reference-precision mode: the previous block's GEMMs consume fp32 operands). */
int eamd_layernorm_bwd_drop_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                const float* dres, float* dx, float* dx_drop, float drop_p, const uint64_t* step_dev,
                                uint64_t salt, float* dgamma, float* dbeta, float* workspace, int rows, int D, void* stream);

/* Masked softmax of attention scores, legacy rel_shift of `bd` fused in.
 * reference: transformer/attention.py:63-90 (mask fill / softmax / zero fill), :141-162 (rel_shift),
 * :200-204 ((ac+bd)/sqrt(d_k)).  Scores are nblocks x [T1, ld] (ld >= T2, pad columns of P are
 * zeroed); block z belongs to batch z % B.  mask (uint8, 0 = masked) element (b,i,j) at
 * mask[b*mask_bstride + i*mask_qstride + j]; NULL = no mask. */
int eamd_softmax_fwd(const float* ac, const float* bd, const unsigned char* mask, int64_t mask_bstride,
                     int64_t mask_qstride, float* P, void* P_bf16, int nblocks, int B, int T1, int T2, int64_t ld,
                     float scale, void* stream);
/* dP is overwritten by d(ac); if dbd != NULL the same values are scattered through the
 * inverse rel_shift. */
/* bf16 variant: P_bf16 in; dS_bf16 (instead of overwriting dP) and dbd_bf16 out.  Every element of dbd is
 * written (scattered value, or zero where the shift never lands and in the pad columns): no pre-zeroing. */
int eamd_softmax_bwd(const float* P, const void* P_bf16, float* dP, float* dbd, void* dS_bf16, void* dbd_bf16,
                     int nblocks, int T1, int T2, int64_t ld, float scale, void* stream);

/* Fused attention forward (bf16 activations, d_k = 64, T2 <= 2048; rows of more than 256 keys run at one workgroup per CU,
 * rows of 513 .. 2048 keys on the 16-query long-row kernels of attn_f32.hip with the operands widened on load): P = softmax(mask(scale * (qu k^T + rel_shift(qv pos^T))))
 * and ctx = P v in one launch; the fp32 scores stay on chip.  reference: transformer/attention.py:63-114 (forward_attention,
 * MultiHeadedAttention.forward), :141-206 (RelPositionMultiHeadedAttention).  All operands are bf16 with heads side by side:
 * element (b, t, h, d) of qu at qu[(b*T1 + t)*ldq + h*64 + d] (k, v: T2 rows per batch; pos: (m, h, d) at pos[m*ldpos + h*64 + d],
 * shared by the batch).  qv / pos are both NULL (no relative positions) or both set (then T1 == T2).  mask as in
 * eamd_softmax_fwd.  Outputs: P_bf16 [H][B][T1][ldp] (pad columns zeroed; kept for the backward pass) and
 * ctx_bf16 (b, t, h, d) at ctx[(b*T1 + t)*ldc + h*64 + d].  EAMD_EUNSUPPORTED for other d_k / longer rows / unaligned
 * operands: callers then run the score GEMMs, eamd_softmax_fwd and the context GEMM instead.
 * Attention dropout (attention.py:91; drop_p > 0): ctx = dropout(P) v with the mask eamd_dropout draws for
 * (drop_step, drop_salt) on the elements of P; P stays undropped (the softmax backward needs it) and the dropped
 * probabilities are written to Pd_bf16 (same shape: the operand of dv = Pd^T dctx). */
/* shift_len (all four entry points; device int32 scalar or NULL = T2): the length the legacy rel_shift
 * (transformer/attention.py:160-171) is taken over.  A batch that a shape-bucketed graph padded beyond its own longest
 * utterance T'max passes T'max: bd is then shifted as the reference's T'max x T'max matrix (the keys from T'max on must be
 * masked by `mask`), so the padded launch reproduces the exact-shape result; read at run time, i.e. per graph replay. */
int eamd_attn_fwd(const void* qu, int64_t ldq, const void* qv, int64_t ldqv, const void* k, int64_t ldk, const void* v,
                  int64_t ldv, const void* pos, int64_t ldpos, const unsigned char* mask, int64_t mask_bstride,
                  int64_t mask_qstride, void* P_bf16, int64_t ldp, void* ctx_bf16, int64_t ldc, int B, int H, int T1,
                  int T2, int dk, float scale, void* Pd_bf16, float drop_p, const uint64_t* drop_step, uint64_t drop_salt,
                  const int32_t* shift_len, void* stream);

/* Query side of the attention backward in one launch (same operand layouts and limits as eamd_attn_fwd):
 * dP = dctx v^T, dS = scale * P (dP - rowsum(P dP)) -> dS_bf16 [H][B][T1][ldp] (pad columns zeroed), its inverse
 * rel_shift scatter -> dbd_bf16 (same shape, every element written; NULL without relative positions, else T1 == T2)
 * and dq = dS k -> dq (fp32 or bf16, element (b, t, h, d) at dq[(b*T1 + t)*ldo + h*64 + d]).  Replaces a score-
 * gradient GEMM, eamd_softmax_bwd and the dq GEMM; the key-side products (dv = P^T dctx, dk = dS^T q, dqv / dpos
 * from dbd) remain GEMMs over P / dS / dbd. */
int eamd_attn_bwd_q(const void* dctx, int64_t ldd, const void* k, int64_t ldk, const void* v, int64_t ldv,
                    const void* P_bf16, int64_t ldp, void* dS_bf16, void* dbd_bf16, void* dq, int64_t ldo, int dq_is_bf16,
                    int B, int H, int T1, int T2, int dk, float scale, float drop_p, const uint64_t* drop_step,
                    uint64_t drop_salt, const int32_t* shift_len, void* stream);   /* drop_*: the forward's attention dropout (dP <- mask * dP / (1 - p)) */

/* fp32 twins of eamd_attn_fwd / eamd_attn_bwd_q (the reference's precision; v_mfma_f32_16x16x4_f32): same operand
 * layouts, limits (d_k = 64, T2 <= 2048 - 513 .. 2048 keys on the long-row kernels -, T1 == T2 with relative positions) and results, every tensor fp32: P, dS, dbd
 * [H][B][T1][ldp] (ldp % 4 == 0, pad columns zeroed), ctx / dq (b, t, h, d).  Row strides are multiples of 4 floats and
 * base pointers 16-byte aligned, else EAMD_EUNSUPPORTED (callers then run the GEMM / eamd_softmax_* path). */
int eamd_attn_fwd_f32(const float* qu, int64_t ldq, const float* qv, int64_t ldqv, const float* k, int64_t ldk,
                      const float* v, int64_t ldv, const float* pos, int64_t ldpos, const unsigned char* mask,
                      int64_t mask_bstride, int64_t mask_qstride, float* P, int64_t ldp, float* ctx, int64_t ldc, int B,
                      int H, int T1, int T2, int dk, float scale, float* Pd, float drop_p, const uint64_t* drop_step,
                      uint64_t drop_salt, const int32_t* shift_len, void* stream);
int eamd_attn_bwd_q_f32(const float* dctx, int64_t ldd, const float* k, int64_t ldk, const float* v, int64_t ldv,
                        const float* P, int64_t ldp, float* dS, float* dbd, float* dq, int64_t ldo, int B, int H, int T1,
                        int T2, int dk, float scale, float drop_p, const uint64_t* drop_step, uint64_t drop_salt,
                        const int32_t* shift_len, void* stream);

/* Key side of the attention backward in one launch (fp32 tensors, d_k = 64; any T1 / T2): dv = Pd^T dctx, dk = dS^T qu and,
 * with relative positions (dbd, qv, dpos all set; T1 == T2), dpos += dbd^T qv summed over the batch (dpos is ACCUMULATED
 * into: positions are shared by the batch; zero it first).  Pd / dS / dbd: [H][B][T1][ldp] as eamd_attn_fwd_f32 /
 * eamd_attn_bwd_q_f32 leave them (Pd = the probabilities the context was built from); dctx / qu / qv: (b, t, h, d) at
 * [(b*T1 + t)*ld + h*64 + d]; dv / dk: (b, j, h, d) at [(b*T2 + j)*ldo + h*64 + d]; dpos: (m, h, d) at [m*ldpos + h*64 + d].
 * Replaces the three batched GEMMs over P / dS / dbd.  reference: autograd of transformer/attention.py:63-114, :141-206. */
int eamd_attn_bwd_kv_f32(const float* Pd, const float* dS, const float* dbd, int64_t ldp, const float* dctx, int64_t ldd,
                         const float* qu, int64_t ldq, const float* qv, int64_t ldqv, float* dv, float* dk_out, int64_t ldo,
                         float* dpos, int64_t ldpos, int B, int H, int T1, int T2, int dk, void* stream);

/* The bf16 twin of the key-side launch (dv and dk only; the positional product stays a split-K GEMM): Pd / dS bf16
 * [H][B][T1][ldp] as eamd_attn_fwd / eamd_attn_bwd_q leave them, dctx / qu bf16 (b, t, h, d), dv / dk bf16 (b, j, h, d)
 * sharing the row stride ldo; fp32 accumulation.  ldp / ldd / ldq multiples of 8, ldo a multiple of 4.
 * reference: autograd of transformer/attention.py:63-114, :141-206. */
int eamd_attn_bwd_kv(const void* Pd_bf16, const void* dS_bf16, int64_t ldp, const void* dctx, int64_t ldd, const void* qu,
                     int64_t ldq, void* dv, void* dk_out, int64_t ldo, int B, int H, int T1, int T2, int dk, void* stream);

/* Label-smoothing KL loss rows + argmax-correct flags + gradient (softmax - true_dist)*inv_denom.
 * reference: transformer/label_smoothing_loss.py:44-63, nets_utils.py:299-319 (th_accuracy). */
int eamd_lsm_loss(const float* logits, const int64_t* target, float* loss_rows, float* correct_rows,
                  float* grad, int rows, int V, int ignore_id, float smoothing, float inv_denom,
                  void* stream);
/* First-max argmax per row (torch.argmax tie-break).  reference: ctc.py:144-151,
 * e2e_asr_transformer.py:274-284 (greedy CTC). */
int eamd_argmax_rows(const float* x, int64_t ld, int32_t* out, int rows, int V, void* stream);
int eamd_reduce_sum(const float* in, int64_t n, float* out, float scale, void* stream);
/* reference: decoder.py:318, ctc.py:134-142. */
int eamd_log_softmax_rows(const float* x, float* y, int rows, int V, void* stream);
/* Bookkeeping of a device-resident beam step after the selection (reference: beam_search.py:177-203, batch_beam_search.py:249-284,
 * there on host objects).  For each of the n = utterances x beam surviving slots s with winner index top_i[s] (= slot * V + token
 * inside its utterance) and score top_s[s]:
 *   hyp_i[s] = the hypothesis it extends, tok_i[s] = the token, pos[s] = the token's position among that hypothesis's candidates
 *              ids [n, ncand] (ids NULL: the token itself);
 *   sc_out [ns, n]: per-scorer scores carried along - rows j < nf from logps[j] [n, V] (HOST array of nf <= 4 device pointers),
 *              row nf (if ns == nf + 1) from c_local [n, ldc] at column token (full_mode) or pos;
 *   yseq_out [n, W] = yseq_in[hyp_i] with position L set to the token;
 *   hyp_out[s] = top_s[s], or -inf when the slot is empty (non-finite score) or ended (token == eos, or maxlen[utterance] <= step + 1);
 *   rec [n, 3 + ns + W] = (step, top_s, token, sc_out[:, s], yseq_out[s, :]) as floats: the step log the host reads. */
int eamd_beam_finish(const float* top_s, const int64_t* top_i, int n, int beam, int V, int W, int L, int step, int eos,
                     const int64_t* maxlen, int ns, int nf, const float* sc_in, const float* const* logps, const float* c_local,
                     int64_t ldc, int full_mode, const int64_t* ids, int ncand, const int64_t* yseq_in, float* sc_out,
                     int64_t* yseq_out, float* hyp_out, int64_t* hyp_i, int64_t* tok_i, int64_t* pos, float* rec, void* stream);
/* nn.Linear on a handful of rows (M <= 16; up to 1024 rows in blocks of 16 with four columns per wave; K a multiple of 4, fp32): y[M,N] = alpha * act(a_act(x) W^T + bias) + R with W [N,K]
 * row-major (nn.Linear layout), act 0 none / 1 relu / 2 swish on the result, a_act an eamd_act id applied to x while it is read.
 * One wave per output column instead of 64-wide tiles walking K alone.  reference: the per-step products of a decoding
 * hypothesis set, transformer/decoder_layer.py:77-134, decoder.py:283-321.  ldx / ldr: row strides of x and R in elements (0 =
 * dense; the newest position of every hypothesis's prefix is a strided set of rows).  EAMD_EUNSUPPORTED: the caller uses eamd_gemm. */
int eamd_linear_rows_f32(const float* x, const float* W, const float* bias, const float* R, float* y, int M, int N, int K,
                         int a_act, int act, float alpha, int64_t ldx, int64_t ldr, void* stream);
/* Cached decoding of the Transformer decoder, csrc/decode.hip (reference: transformer/decoder.py:283-321, decoder_layer.py:81-134).
 * eamd_linear_rows_ln_f32: y[M, N] (row stride ldy) = alpha * act(LayerNorm(x; gamma, beta, eps) W^T + bias) + R for M <= 16 rows,
 *   K <= 1024 (or 16 < M <= 1024 rows at K <= 256, K % 16 == 0: 16-row blocks on the matrix cores): the pre-norm of a decoder sub-block inside the product behind it (layer_norm.py:12-38 in front of attention.py:40-61 /
 *   positionwise_feed_forward.py:28 / decoder.py:312-317); row strides 0 = dense.
 * eamd_decode_self_attn: self-attention of the NEWEST position of n hypotheses over their prefixes, keys / values cached per layer in
 *   time-major [Lcap, n, D] buffers: row (pos, slot) is written from qkv [n, ldq] = (q | k | v) of this step, rows t < pos are read
 *   at slot_at[slot][t] ([n, Lcap] int32: the slot that held this hypothesis's ancestor at position t - a beam step re-orders that
 *   table, never the caches).  d_k = 64 (D = 64 H); ctx [n, D]; no mask (a prefix has no padding).  The reference re-projects keys and
 *   values of the whole prefix at every step from cached layer outputs (decoder_layer.py:88-107): same numbers.
 * eamd_beam_slots: slot_out[i][t] = slot_in[hyp[i]][t] (t < pos), slot_out[i][pos] = hyp[i] - the table behind a selection. */
int eamd_linear_rows_ln_f32(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                            const float* R, float* y, int M, int N, int K, int act, float alpha, int64_t ldx, int64_t ldr,
                            int64_t ldy, void* stream);
int eamd_decode_self_attn(const float* qkv, int64_t ldq, float* kcache, float* vcache, const int32_t* slot_at, int Lcap, int pos,
                          int n, int H, int D, float* ctx, void* stream);
int eamd_beam_slots(const int32_t* slot_in, int32_t* slot_out, const int64_t* hyp, int n, int Lcap, int pos, void* stream);
/* Source attention of a beam step (decoder_layer.py:109-121 on one query position per hypothesis): hypothesis r of nutt * g belongs to
 * utterance r / g and attends over that utterance's T memory frames; kmem / vmem address the keys / values of ONE layer inside the
 * decoder stack's shared projection of the memory ([nutt, T, ldkv] row stride ldkv); mask [nutt, T] uint8 (0 = padded frame, may be
 * NULL); q [nutt * g, ldq]; ctx [nutt * g, D]; d_k = 64.  Masked frames get probability 0, an all-masked row gives zeros
 * (attention.py:80-88). */
int eamd_decode_src_attn(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv, const uint8_t* mask, int nutt,
                         int g, int T, int H, int D, float* ctx, void* stream);
/* ... one workgroup per (utterance, head) for all g <= 16 hypotheses of the utterance (many utterances per search: the keys and
 * values of an utterance are read once per head, not once per hypothesis); T <= 1024. */
int eamd_decode_src_attn_group(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv, const uint8_t* mask, int nutt,
                         int g, int T, int H, int D, float* ctx, void* stream);
/* ... with the keys of every utterance split over `splits` (2 .. 16) workgroups and a merge launch behind them (row maxima, sums and
 * unnormalised partial contexts meet in ws: eamd_decode_src_attn_split_workspace floats); T <= 4096. */
int64_t eamd_decode_src_attn_split_workspace(int nutt, int g, int H, int splits);
int eamd_decode_src_attn_split(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv, const uint8_t* mask,
                               int nutt, int g, int T, int H, int D, int splits, float* ws, float* ctx, void* stream);
/* The selection of a beam step on the pre-beam candidates (reference: beam_search.py:296-334 with :199-226: tokens outside the pre-beam
 * are dropped, so an utterance's `beam` best continuations are among its beam x P candidates).
 * eamd_weighted_sum: out[i] = ((0 + w_0 logp_0[i]) + w_1 logp_1[i]) + ... over nf <= 4 full scorers ([n, V] each; numel = n V, a
 *   multiple of 4): the sum the pre-beam top-k is taken on (beam_search.py:298-309), in the reference's order of operations.
 * eamd_beam_select: per utterance (nutt of them, `beam` slots each), candidate (slot, j) scores
 *   (pre[slot][ids[slot][j]] + w_ctc (psi[slot][j] - c_s[slot])) + hyp[slot]; the best `beam` leave as top_s / top_i [nutt, beam]
 *   (top_i = local slot * V + token; value descending, ties by ascending top_i = torch.topk on the flattened scores; NaN ranks as
 *   -inf), c_local [n, P] = psi - c_s (the partial scorer's score of each candidate, for eamd_beam_finish).  beam * P <= 1024. */
int eamd_weighted_sum(const float* const* logps, const float* weights, int nf, int64_t numel, float* out, void* stream);
int eamd_beam_select(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc,
                     int nutt, int beam, int P, int V, float* c_local, float* top_s, int64_t* top_i, void* stream);
/* The k (<= 64) largest of each row of x [rows, n] (row stride ld), sorted by value descending, equal values by ascending index;
 * NaN counts as -inf.  vals / idx [rows, k].  reference: the torch.topk selections of a beam step (beam_search.py:143-176,
 * batch_beam_search.py:86-110: pre-beam over V, best `beam` of beam x V). */
int eamd_topk_rows(const float* x, int64_t ld, int rows, int n, int k, float* vals, int64_t* idx, void* stream);
/* ... the indices also as int32 (idx32 [rows, k], may be NULL): the candidate list eamd_ctc_prefix_psi takes. */
int eamd_topk_rows_i32(const float* x, int64_t ld, int rows, int n, int k, float* vals, int64_t* idx, int32_t* idx32, void* stream);
/* One hipGraph for EVERY beam step (reference: the step index `i` of beam_search.py:349-364's loop, here a device integer): the
 * variants below read the step-dependent integer from device memory - value = *dev + the host argument, which becomes an offset -
 * so a captured step does not bake it in.  eamd_beam_step_dyn reads step (L = step + 1) from step_dev, writes the log row into
 * slot step % ring of a [ring][n][3 + ns + W] ring (ring 0: as before) and leaves step + 1 in step_out; with slot_in / slot_out
 * ([n, Lcap] int32) it also does eamd_beam_slots' re-ordering of the cached decoder's slot table (position = step).  eamd_copy_jobs: up to 16
 * small device-to-device copies (sizes in bytes, multiples of 4) in one launch - the state a step hands to the next replay. */
int eamd_decode_self_attn_dyn(const float* qkv, int64_t ldq, float* kcache, float* vcache, const int32_t* slot_at, int Lcap, int pos,
                              const int32_t* pos_dev, int n, int H, int D, float* ctx, void* stream);
int eamd_beam_slots_dyn(const int32_t* slot_in, int32_t* slot_out, const int64_t* hyp, int n, int Lcap, int pos, const int32_t* pos_dev,
                        void* stream);
int eamd_beam_step_dyn(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc, int nutt,
                       int beam, int P, int V, int W, int L, int step, int eos, const int64_t* maxlen, int ns, int nf, const float* sc_in,
                       const float* const* logps, const int64_t* yseq_in, float* c_local, float* sc_out, int64_t* yseq_out, float* hyp_out,
                       int64_t* hyp_i, int64_t* tok_i, int32_t* tok32, float* cs_out, float* rec, const int32_t* step_dev, int32_t* step_out,
                       int ring, const int32_t* slot_in, int32_t* slot_out, int Lcap, void* stream);
int eamd_ctc_prefix_psi_dyn(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev, const int32_t* cand,
                            const int32_t* last, int olen, const int32_t* olen_dev, float* psi, int ncand, int Tmax, int V, int blank,
                            int eos, void* stream);
int eamd_ctc_prefix_state_dyn(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev, const int64_t* parent,
                              const int64_t* tok, const int32_t* last, int olen, const int32_t* olen_dev, const float* alive, float* r_out,
                              int Tmax, int V, int blank, void* stream);
int eamd_embed_pe_dyn(const int64_t* tok, int64_t ldt, const float* table, const float* pe, float* out, int64_t rows, int U, int D,
                      float scale, int pos_offset, const int32_t* pos_dev, void* stream);
int eamd_copy_jobs(const void* const* src, void* const* dst, const int64_t* nbytes, int njobs, void* stream);
/* eamd_weighted_sum + eamd_topk_rows_i32 in one launch: pre [rows, n] = sum_j weights[j] * logps[j] (written out; HOST arrays of
 * nf <= 4 device pointers / floats, the same separately rounded arithmetic), and the k largest of each of its rows.
 * extra >= 0: vals / idx / idx32 are [rows, k + 1], the last column = the token `extra`, or -1 where it is already among the k
 * (the <eos> a "full"-mode partial scorer scores besides the pre-beam: batch_beam_search.py:221-231, scorers/ctc.py:82-96). */
int eamd_weighted_topk_rows(const float* const* logps, const float* weights, int nf, int rows, int n, int k, int extra, float* pre,
                            float* vals, int64_t* idx, int32_t* idx32, void* stream);
/* eamd_beam_select + eamd_beam_finish of a BeamSearch step with a pre-beam in one launch (one workgroup per utterance; same
 * arithmetic): the arguments of both - ids [n, P] the candidates, ns == nf + 1 (the partial scorer's row is last), W the width of the
 * prefix buffers, L the position the new token takes - plus tok32 [n] = tok_i as int32 (the next step's `last`) and
 * cs_out [n] = psi at the chosen candidate of the extended hypothesis (the partial scorer's running prefix score).  A candidate
 * id < 0 is no candidate (it is never selected).  beam <= 64,
 * beam * P <= 1023, beam * V < 2^31.  reference: beam_search.py:143-226,296-334. */
int eamd_beam_step(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc, int nutt,
                   int beam, int P, int V, int W, int L, int step, int eos, const int64_t* maxlen, int ns, int nf, const float* sc_in,
                   const float* const* logps, const int64_t* yseq_in, float* c_local, float* sc_out, int64_t* yseq_out, float* hyp_out,
                   int64_t* hyp_i, int64_t* tok_i, int32_t* tok32, float* cs_out, float* rec, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise helpers.
 * ------------------------------------------------------------------------------------------ */
int eamd_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, void* stream);
/* fp32 -> bf16 (RNE) copy: bf16 shadows of weights / activations that feed the bf16-operand GEMM */
int eamd_cast_bf16(const float* x, void* y_bf16, int64_t n, void* stream);
int eamd_scale_dev(const float* x, const float* scale_dev, float* out, int64_t n, float extra, void* stream);
int eamd_act_fwd(const float* x, float* y, int64_t n, int act, void* stream);
int eamd_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, void* stream);
/* reference: conformer/convolution.py:72 (GLU). x is [rows, 2C], y/dy [rows, C]. */
int eamd_glu_fwd(const float* x, float* y, int64_t rows, int C, void* stream);
/* dx (fp32) or dx_bf16 (bf16, when non-NULL) receives the [rows, 2C] gradient */
int eamd_glu_bwd(const float* dy, const float* x, float* dx, void* dx_bf16, int64_t rows, int C, void* stream);
/* reference: transformer/attention.py:186-190 (q + pos_bias_u, q + pos_bias_v). */
/* bf16 = 1: q, qu, qv are bf16.  q has row stride ldq (a column block of the fused QKV projection); qu, qv dense */
int eamd_add_bias2(const void* q, int64_t ldq, const float* u, const float* v, void* qu, void* qv, int64_t rows, int D,
                   int bf16, void* stream);
/* out_bf16[r * ld_out + c] = a[r, c] + b[r, c] (b optional), dense fp32 [rows, cols] inputs */
int eamd_add_cast_bf16(const float* a, const float* b, void* out_bf16, int64_t rows, int cols, int64_t ld_out,
                       void* stream);
/* fp32 twin: out[r * ld_out + c] = a[r, c] + b[r, c] (b may be NULL) into a column block of a wider fp32 matrix
 * (dq = dqu + dqv into the fused [rows, 3D] q/k/v gradient in fp32 mode). */
int eamd_add_block_f32(const float* a, const float* b, float* out, int64_t rows, int cols, int64_t ld_out, void* stream);
/* out_bf16[r * ld_out + c] = a[r, c] + b[r, c] and, from the same pass, suma[c] += sum_r a[r, c], sumb[c] += sum_r b[r, c]
 * (dense fp32 [rows, D] inputs, D even and <= 512, else EAMD_EUNSUPPORTED): dq = dqu + dqv and the gradients of
 * pos_bias_u / pos_bias_v in the backward of attention.py:186-190. */
int eamd_add_cast_colsum2(const float* a, const float* b, void* out_bf16, int64_t ld_out, float* suma, float* sumb,
                          int64_t rows, int D, void* stream);
/* fp32 twin (reference-precision mode): out[r*ld_out + c] = a + b as fp32, same column sums. */
int eamd_add_colsum2_f32(const float* a, const float* b, float* out, int64_t ld_out, float* suma, float* sumb, int64_t rows,
                         int D, void* stream);
/* out[D] += scale * column sums of x[rows, D] (bias gradients). */
int eamd_colsum(const void* x, int64_t ld, float* out, int64_t rows, int D, float scale, int x_bf16, void* stream);
/* reference: decoder.py:83-86,251 (Embedding + PositionalEncoding), embedding.py:80-91.
 * pe may be NULL (plain nn.Embedding lookup: rnn/decoders.py:88, transducer/rnn_decoder.py:44);
 * pad_idx >= 0 in the backward = nn.Embedding(padding_idx): that row gets no gradient (-1: none). */
int eamd_embed_pe(const int64_t* tok, const float* table, const float* pe, float* out, int64_t rows, int U,
                  int D, float scale, int pos_offset, void* stream);
/* ... token r read at tok[r * ldt] (the newest column of a [n, W] prefix buffer: a beam step's input). */
int eamd_embed_pe_ld(const int64_t* tok, int64_t ldt, const float* table, const float* pe, float* out, int64_t rows, int U,
                     int D, float scale, int pos_offset, void* stream);
int eamd_embed_bwd(const int64_t* tok, const float* dout, float* dtable, int64_t rows, int D, float scale,
                   int64_t pad_idx, void* stream);
int eamd_posenc(const float* x, const float* pe, float* out, int64_t rows, int T, int D, float scale,
                void* stream);
/* ScaledPositionalEncoding (transformer/embedding.py:95-128): out = x * scale + alpha[0] * pe[t] with the learnable
 * scalar alpha read on the device (no host sync); the backward entry accumulates dalpha += sum(dout * pe[t]) (dx is
 * dout * scale, an eamd_axpby). */
int eamd_posenc_scaled(const float* x, const float* pe, const float* alpha, float* out, int64_t rows, int T, int D,
                       float scale, void* stream);
int eamd_posenc_scaled_bwd(const float* dout, const float* pe, float* dalpha, int64_t rows, int T, int D,
                           void* stream);
int eamd_permute4(const float* src, float* dst, int d0, int d1, int d2, int d3, int64_t s0, int64_t s1,
                  int64_t s2, int64_t s3, int accumulate, void* stream);
/* y = act(x) * keep / (1-p), keep(i) = hash(step_dev[0], salt, i) >= p; the same call on a gradient applies the
 * same mask (backward).  step_dev: device counter advanced once per training step (graph-replay safe).
 * reference: nn.Dropout call sites (conformer/encoder_layer.py:55, positionwise_feed_forward.py:27, ctc.py:85). */
int eamd_dropout(const void* x, void* y, int64_t n, float p, const uint64_t* step_dev, uint64_t salt, int act,
                 int in_bf16, int out_bf16, void* stream);
int eamd_rng_advance(uint64_t* step_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conformer convolution module + first subsampling convolution (channels-last activations).
 * reference: conformer/convolution.py:13-79, transformer/subsampling.py:28-33.
 * ------------------------------------------------------------------------------------------ */
int eamd_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int K,
                    void* stream);
int eamd_dwconv_bwd_x(const float* dy, const float* w, float* dx, int B, int T, int C, int K, void* stream);
/* dw[C,K], db[C] are ACCUMULATED. */
int eamd_dwconv_bwd_w(const float* dy, const float* x, float* dw, float* db, int B, int T, int C, int K,
                      void* stream);
/* GLU-fused twins for the Conformer convolution module (conformer/convolution.py:53-79): `a` = pointwise_conv1 output
 * [B, T, 2C] (value columns | gate columns).  eamd_dwconv_glu_fwd: y = dwconv(GLU(a)) without writing GLU(a);
 * eamd_dwconv_glu_bwd_w: dw / db += as eamd_dwconv_bwd_w with x = GLU(a) formed on load; eamd_dwconv_glu_bwd_x:
 * da [B, T, 2C] (fp32, or bf16 when da_bf16) = GLU'(a) . dwconv_bwd_x(dy) - the depthwise input gradient never written. */
int eamd_dwconv_glu_fwd(const float* a, const float* w, const float* bias, float* y, float* bn_part, int B, int T, int C, int K,
                        void* stream);   /* bn_part (optional): 3*C*B*ceil(T/64) floats of BatchNorm partial statistics of y */
int eamd_dwconv_glu_bwd_x(const float* dy, const float* w, const float* a, void* da, int da_bf16, int B, int T, int C, int K,
                          void* stream);
int eamd_dwconv_glu_bwd_w(const float* dy, const float* a, float* dw, float* db, int B, int T, int C, int K, void* stream);
int eamd_bn_nslab(int64_t M, int C);
/* BatchNorm1d training statistics over [M, C]; workspace 3*C*nslab floats; running stats updated in
 * place (momentum, unbiased variance) and num_batches_tracked[0] += 1 when non-NULL (torch.nn.BatchNorm1d's buffers). */
int eamd_bn_stats(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                  float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float eps, float momentum, void* stream);
/* Second stage alone: merges nslab slabs of [count | mean | M2] x C (eamd_dwconv_glu_fwd's bn_part) into mean / rstd and
 * updates the running statistics and the batch counter as eamd_bn_stats does. */
int eamd_bn_finalize(const float* part, int nslab, float* mean, float* rstd, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, int C, float eps, float momentum, void* stream);
int eamd_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  void* y, int64_t M, int C, int act, int y_bf16, void* stream);
/* Time-bounded variants for batches that a shape-bucketed graph padded beyond their own longest utterance (rows are
 * (b, t) with t = row % T; bound = device int32 scalar, read at run time): rows with t >= bound[0] take no part in the
 * statistics / sums, count as B * bound rows, and receive dx = 0 - what conformer/convolution.py:56-79 computes on the
 * batch cropped to its own length.  eamd_mask_time zeroes those rows of an [rows, C] tensor in place (the depthwise
 * convolution then sees the zero padding the reference has there). */
int eamd_bn_stats_bounded(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float eps, float momentum, int T,
                          const int32_t* bound, void* stream);
int eamd_bn_bwd_bounded(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                        int act, int training, int T, const int32_t* bound, void* stream);
int eamd_mask_time(void* x, int64_t rows, int C, int T, const int32_t* bound, int is_bf16, void* stream);
/* workspace (2*nslab+2)*C floats; dgamma/dbeta ACCUMULATED. */
int eamd_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                int act, int training, void* stream);
/* Conv2d(1, C, 3, 2) + ReLU, x [B,T,F] -> y [B,H,W,C] (NHWC). */
int eamd_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                   int y_bf16, void* stream);
/* weight / bias gradient (accumulated; dy already ReLU-masked).  workspace: eamd_conv1_bwd_w_workspace(B,T,C) floats
 * of scratch for the two-stage reduction (NULL => f32 atomics straight into dw / db). */
int64_t eamd_conv1_bwd_w_workspace(int B, int T, int C);
int eamd_conv1_bwd_w(const void* dy, const float* x, float* dw, float* db, float* workspace, int B, int T, int F,
                     int C, int dy_bf16, void* stream);

/* ------------------------------------------------------------------------------------------
 * CTC loss (warp-ctc operator slot).  reference: ctc.py:30-47,53-66 (ctc_type warpctc|builtin),
 * espnet2/asr/ctc.py:32-66.  acts are RAW activations (softmax fused), element strides select
 * (T,B,V) or (B,T,V) layout.  nll[b] = -log p(y_b|x_b) (+inf if infeasible);
 * grad = grad_scale * d(sum_b nll_b)/d(acts), zero for t >= ilens[b].
 * ------------------------------------------------------------------------------------------ */
int64_t eamd_ctc_workspace_bytes(int B, int T, int Lmax);
int eamd_ctc_loss(const float* acts, int64_t stride_t, int64_t stride_b, const int64_t* ys_pad,
                  const int32_t* ilens, float* nll, float* grad, int64_t gstride_t, int64_t gstride_b,
                  void* workspace, int B, int T, int V, int Lmax, int blank, int ignore_id, float grad_scale,
                  void* stream);

/* CTC prefix scores of (hypothesis, candidate) pairs for joint CTC/attention beam search.
 * reference: espnet/nets/ctc_prefix_score.py:224-310, scorers/ctc.py:11-127.
 * logp [T,V]; r_prev [nhyp,T,2]; cand [nhyp,ncand]; last/olen [nhyp]; psi [nhyp,ncand]; r_new [nhyp,ncand,T,2]. */
int eamd_ctc_prefix_score(const float* logp, const float* r_prev, const int32_t* cand, const int32_t* last,
                          const int32_t* olen, float* psi, float* r_new, int nhyp, int ncand, int T, int V,
                          int blank, int eos, void* stream);

/* The same for the hypotheses of `nutt` utterances in one launch (batched beam search): hypothesis h belongs to utterance
 * h / per_utt; logp [nutt, Tmax, V] with lens[u] valid frames; r_prev [nutt*per_utt, Tmax, 2], r_new [.., ncand, Tmax, 2]
 * (rows from lens[u] on are neither read nor written).  reference: ctc_prefix_score.py:12-221 (CTCPrefixScoreTH). */
int eamd_ctc_prefix_score_batch(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                const int32_t* cand, const int32_t* last, const int32_t* olen, float* psi, float* r_new,
                                int ncand, int Tmax, int V, int blank, int eos, void* stream);
/* The same scores with the serial recursion OFF a beam step's critical path (csrc/ctc.hip): log psi of a candidate is a logsumexp over
 * the frames of phi(t-1) + x(t) (ctc_prefix_score.py:290-296 never reads r[t] for it) - eamd_ctc_prefix_psi forms it as a parallel
 * reduction (one wave per (hypothesis, candidate); Tmax <= 2048) - and only the continuations that SURVIVE the selection need their
 * forward variables: eamd_ctc_prefix_state runs the recursion of :291-295 for slot s = (hypothesis parent[s], token tok[s]) into
 * r_out [n, Tmax, 2] (slots with alive[s] = -inf get log-zero rows); the caller issues it at the start of the next step on a second
 * stream beside the decoder stack.  olen = prefix length - 1 of the scored hypotheses (one value: all hypotheses of a step have the
 * same length); last [n] = their last tokens.  Same values as eamd_ctc_prefix_score_batch up to the order of the log-sum-exp. */
int eamd_ctc_prefix_psi(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev, const int32_t* cand,
                        const int32_t* last, int olen, float* psi, int ncand, int Tmax, int V, int blank, int eos, void* stream);
int eamd_ctc_prefix_state(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev, const int64_t* parent,
                          const int64_t* tok, const int32_t* last, int olen, const float* alive, float* r_out, int Tmax, int V, int blank,
                          void* stream);

/* ---- feature-side layers ----------------------------------------------
 * SpecAugment on a [B,T,F] batch (x != y).  reference: espnet2/asr/specaug/specaug.py:19-84,
 * espnet2/layers/time_warp.py:15-94 (bicubic F.interpolate along time, align_corners=False, of the segments left and
 * right of `center` to lengths `warped` and len - warped), espnet2/layers/mask_along_axis.py:7-62 (zero fill).
 * lens[B] (NULL = T): frames >= lens[b] become 0 (pad_list of the per-utterance warp path);
 * center/warped[B] (NULL or center < 0 = no warp); fpos/flen [B,nf], tpos/tlen [B,nt]: mask pos <= i < pos + len. */
int eamd_specaug(const float* x, float* y, const int32_t* lens, const int32_t* center, const int32_t* warped,
                 const int32_t* fpos, const int32_t* flen, int nf, const int32_t* tpos, const int32_t* tlen, int nt, int B,
                 int T, int F, void* stream);
/* GlobalMVN: y = ((x - mean[f]) with padded frames zeroed) / std[f]; mean / std / lens each optional.
 * reference: espnet2/layers/global_mvn.py:62-98 */
int eamd_global_mvn(const float* x, float* y, const int32_t* lens, const float* mean, const float* stdv, int B, int T,
                    int F, void* stream);
/* UtteranceMVN, statistics over the valid frames of each utterance; follows utterance_mvn.py:62-88 literally
 * (norm_means leaves -mean in the padding; with both flags the divisor is sqrt(clamp(sqrt(var), eps))).
 * workspace: 2*B*F floats. */
int eamd_utterance_mvn(const float* x, float* y, const int32_t* lens, float* workspace, int norm_means, int norm_vars,
                       float eps, int B, int T, int F, void* stream);

/* ---- log-mel frontend ------------------------------------------------
 * reference: espnet2/layers/stft.py:62-111 (torch.stft: hann window, center=True -> reflect padding of the padded
 * batch by n_fft/2, hop_length, onesided), espnet2/asr/frontend/default.py:93-133 (power spectrum),
 * espnet2/layers/log_mel.py:20-75 (librosa mel matrix, clamp 1e-10, log, padded frames zeroed).
 * The DFT itself is an eamd_gemm call in fp32: A = the padded waveform read as overlapping rows (lda = hop),
 * B = the windowed DFT basis [2F, n_fft] with interleaved (cos, -sin) rows.
 * eamd_reflect_pad: y[b, j] = x[b, reflect(j - pad)] for j < L + 2 pad, 0 for the rest of the row (ldy >= L + 2 pad). */
int eamd_reflect_pad(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int L, int pad, void* stream);
/* spec: frame t of utterance b = row b*rows_per_utt + t, F interleaved (re, im) pairs, row stride ld (even) -
 * or, with power_input != 0, F power values per row (LogMel on its own); melmat [F, M]; lo/hi [M] = non-zero bin range of each mel filter; flens [B] (NULL = T) valid frames;
 * out[b,t,m] = log(max(sum_f |spec|^2 melmat[f,m], 1e-10)) * log_scale, 0 for t >= flens[b]. */
int eamd_logmel(const float* spec, int64_t ld, int64_t rows_per_utt, const float* melmat, const int32_t* lo,
                const int32_t* hi, const int32_t* flens, float* out, int B, int T, int F, int M, float log_scale,
                int power_input, void* stream);

/* ---- recurrent layers (RNN paths) ------------------------------
 * One LSTM step on gate pre-activations gates[B,4H] = x W_ih^T + b_ih + h W_hh^T + b_hh (the products are
 * eamd_gemm calls), gate order i,f,g,o as torch.nn.LSTM / LSTMCell.
 * reference: rnn/encoders.py:15-162 (torch.nn.LSTM on packed sequences), rnn/decoders.py:88-101,120-134,
 * transducer/rnn_decoder.py:47-57,106-138 (LSTMCell stacks).
 * live[B] (optional, uint8): 0 = this sequence has ended (pack_padded_sequence semantics): the state is
 * carried through unchanged (needs h_prev) and the output row y is 0.  acts[B,4H] keeps the activated
 * gates for the backward. */
int eamd_lstm_cell_fwd(const float* gates, const float* c_prev, const float* h_prev, const uint8_t* live, float* h,
                       float* c, float* y, float* acts, int B, int H, void* stream);
/* dy = gradient wrt the step output y, dh = gradient wrt h from the next step, dc = gradient wrt c
 * (each may be NULL = 0).  dgates[B,4H]: pre-activation gradients; dh_pass: share of dh that flows
 * unchanged to h_prev (rows with live == 0), required when live is given. */
int eamd_lstm_cell_bwd(const float* dy, const float* dh, const float* dc, const float* acts, const float* c_prev,
                       const float* c, const uint8_t* live, float* dgates, float* dc_prev, float* dh_pass, int B, int H,
                       void* stream);
/* One LSTM time step as ONE launch: recurrent product h_prev W_hh^T (fp32 MFMA, both operands read as fragments from
 * L2) + gx_t + b_hh + cell update; the backward twin computes dh = dh_pass_in + dgates_next W_hh (w_t = W_hh^T) and the
 * cell backward.  H % 64 == 0 and B <= 64, else EAMD_EUNSUPPORTED (use eamd_gemm + eamd_lstm_cell_*).
 * reference: torch.nn.LSTM / LSTMCell steps, rnn/encoders.py:36-117, rnn/decoders.py:120-134. */
int eamd_lstm_step_fwd(const float* gx, const float* w_hh, const float* b_hh, const float* h_prev, const float* c_prev,
                       const uint8_t* live, float* h, float* c, float* y, float* acts, int B, int H, void* stream);
int eamd_lstm_step_bwd(const float* dy, const float* dgates_next, const float* w_t, const float* dh_pass_in, const float* dc,
                       const float* acts, const float* c_prev, const float* c, const uint8_t* live, float* dgates,
                       float* dc_prev, float* dh_pass, int B, int H, void* stream);
/* A whole LSTM sequence - all T time steps of one or two independent recurrences (the two directions of a BLSTM
 * layer) - in ONE persistent launch: recurrent weights and cell state stay in registers, h_t (forward) / dgates_t
 * (backward) are handed from workgroup to workgroup through write-through stores + one flag per workgroup, both
 * directions run side by side (csrc/lstm_seq.hip).  Layouts as the step entry points: gx / acts / dgates [T, B, 4H],
 * h_out / c_out / y / dy [T, B, H], live [T, B] (0 = padding frame of a packed sequence) or NULL, initial states 0;
 * `reverse` runs the recurrence from t = T-1 down (the backward of a job walks against its forward direction).
 * sync_ws: eamd_lstm_seq_sync_bytes() bytes of device memory at the start of an allocation, zeroed by the call (a small
 * kernel ahead of the persistent one); eamd_lstm_seq_status() copies its status word to the host: 0 = every wait completed
 * (a wait that gives up after 3 s also writes NaN into that step's outputs, so a training step cannot pass silently; do not run two
 * such launches concurrently on one device - each needs one workgroup per CU resident).
 * EAMD_EUNSUPPORTED when the shape does not fit (H % 64, B <= 64, one workgroup per CU for all jobs): run the jobs one
 * per call, or the per-step entry points.
 * reference: rnn/encoders.py:36-39,110-117 (torch.nn.LSTM, bidirectional, packed sequences). */
typedef struct {
  const float* gx;
  const float* w_hh;     /* [4H, H] */
  const float* b_hh;     /* [4H] or NULL */
  const uint8_t* live;
  float* h_out;
  float* c_out;
  float* y;              /* h with padding frames zeroed, or NULL */
  float* acts;           /* the four gate activations per (t, b, unit), for backward */
  int32_t reverse;
  int32_t reserved;
} eamd_lstm_seq_fwd_t;
typedef struct {
  const float* dy;       /* gradient wrt y, or NULL */
  const float* w_t;      /* W_hh^T [H, 4H] */
  const float* acts;
  const float* c_out;
  const uint8_t* live;
  float* dgates;         /* out: gate pre-activation gradients */
  int32_t reverse;       /* the job's FORWARD direction */
  int32_t reserved;
} eamd_lstm_seq_bwd_t;
int64_t eamd_lstm_seq_sync_bytes(void);
int eamd_lstm_seq_fwd(const eamd_lstm_seq_fwd_t* jobs, int njobs, int T, int B, int H, void* sync_ws, void* stream);
int eamd_lstm_seq_bwd(const eamd_lstm_seq_bwd_t* jobs, int njobs, int T, int B, int H, void* sync_ws, void* stream);
int eamd_lstm_seq_status(const void* sync_ws, void* stream);
/* sticky[0] (int32, device, caller-owned, cleared by the caller) takes the status word of the launch that used sync_ws unless it
 * already holds an earlier failure: one tiny launch behind each persistent launch, no synchronisation.  Replaces reading
 * eamd_lstm_seq_status per launch in a training loop (trainer.py:439-455 only sees a non-finite gradient norm). */
int eamd_lstm_seq_status_merge(const void* sync_ws, void* sticky, void* stream);
/* One GRU step (torch.nn.GRU / GRUCell, gate order r,z,n; rnn/encoders.py:31-33,110-119, rnn/decoders.py:96,105,
 * transducer/rnn_decoder.py:50) on gx = x W_ih^T + b_ih and gh = h W_hh^T + b_hh (both [B,3H], eamd_gemm products).
 * acts [B,4H] = r, z, n, gh_n.  Backward: dgx, dgh [B,3H] and dh_direct [B,H] (the part of the gradient that reaches
 * h_prev directly: z * dh', or all of it on rows with live == 0). */
int eamd_gru_cell_fwd(const float* gx, const float* gh, const float* h_prev, const uint8_t* live, float* h, float* y,
                      float* acts, int B, int H, void* stream);
int eamd_gru_cell_bwd(const float* dy, const float* dh, const float* acts, const float* h_prev, const uint8_t* live,
                      float* dgx, float* dgh, float* dh_direct, int B, int H, void* stream);
/* F.max_pool2d(x, 2, stride=2, ceil_mode=True) on NHWC activations (VGG2L, rnn/encoders.py:205,208).
 * idx[B,Ho,Wo,C] keeps the position (0..3) of the maximum inside its window. */
int eamd_maxpool2x2_fwd(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, void* stream);
int eamd_maxpool2x2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C, void* stream);
/* y[r,:] = keep[r] ? x[r,:] : 0  (zeroing of padded encoder frames, rnn/encoders.py:323-325; x may alias y) */
int eamd_mask_rows(const float* x, const uint8_t* keep, float* y, int64_t rows, int D, void* stream);
/* conv1d positionwise layers (MultiLayeredConv1d / Conv1dLinear, transformer/multi_layer_conv.py:13-105): im2col along
 * time so that Conv1d(C, N, k, padding=(k-1)/2) over [B,T,C] is one eamd_gemm against the [N, k*C] tap-major weight.
 * col[(b,t), kk*C + c] = x[b, t+kk-(k-1)/2, c] (zero outside 0 <= . < T); x / col fp32 or both bf16 (C even);
 * eamd_fold1d is the adjoint on fp32: dx[b,t,c] = sum_kk dcol[(b, t-kk+(k-1)/2), kk*C + c]. */
int eamd_unfold1d(const void* x, void* col, int B, int T, int C, int k, int bf16, void* stream);
int eamd_fold1d(const float* dcol, float* dx, int B, int T, int C, int k, void* stream);
/* VGG2L first convolution (1 -> C channels, 3x3, stride 1, padding 1) + ReLU on [B,T,F] -> NHWC [B,T,F,C], and
 * its weight gradient (dy already ReLU-masked).  reference: rnn/encoders.py:184,203. */
int eamd_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                        int y_bf16, void* stream);
int64_t eamd_conv3x3_c1_bwd_w_workspace(int B, int T, int C);
int eamd_conv3x3_c1_bwd_w(const void* dy, const float* x, float* dw, float* db, float* workspace, int B, int T, int F,
                          int C, int dy_bf16, void* stream);
/* Location-aware attention, one decoder step.  reference: rnn/attentions.py:300-380 (AttLoc.forward).
 *   conv = Conv2d(1,C,(1,K))(att_prev) (K = 2*aconv_filts+1, no bias); e = gvec . tanh(W_att conv + pre_enc +
 *   dec_proj) + gb, -inf for t >= lens[b]; w = softmax(scaling * e); ctx = sum_t w * enc_h.
 * pre_enc = mlp_enc(enc_h) [B,T,A] and dec_proj = mlp_dec(dec_z) [B,A] are eamd_gemm products.
 * th [B,T,A] (tanh output) and conv [B,T,C] are kept for the backward.
 * C = 0 (att_prev, conv_w, w_att, conv NULL): no location term = additive attention (AttAdd, attentions.py:167-247,
 * and the per-head energies of AttMultiHeadAdd :993-1107).
 * R = rows of attention history the convolution spans: 1 for AttLoc; att_win for AttLoc2D (attentions.py:485-603),
 * where att_prev is [B,R,T] and conv_w [C,1,R,K]. */
int eamd_attloc_fwd(const float* att_prev, const float* conv_w, const float* w_att, const float* pre_enc,
                    const float* dec_proj, const float* gvec, const float* gb, const int32_t* lens, const float* enc_h,
                    float scaling, float* e, float* th, float* conv, float* w, float* ctx, int B, int T, int A, int C,
                    int K, int R, int E, void* stream);
/* backward stage 1: from d ctx [B,E] and the gradient arriving at w from the next step (dw_ext, may be NULL):
 * de [B,T], d_enc_h [B,T,E] (= w * dctx), df [B,T,A] (gradient at the tanh input = d pre_enc);
 * dgvec [A], dgb [1], d_dec_proj [B,A] are ACCUMULATED. */
int eamd_attloc_bwd_energy(const float* dctx, const float* dw_ext, const float* w, const float* enc_h, const float* th,
                           const float* gvec, float scaling, float* de, float* d_enc_h, float* df, float* dgvec,
                           float* dgb, float* d_dec_proj, int B, int T, int A, int E, void* stream);
/* Stage 1 with the two products over mlp_att's weight folded in (AttLoc with its location convolution, C <= 16 channels,
 * A <= 1024, A % 4 == 0): everything eamd_attloc_bwd_energy does, plus dconv [B,T,C] = df @ W_att (written) and
 * dw_att [A,C] += df^T conv (ACCUMULATED), with conv [B,T,C] as eamd_attloc_fwd left it and w_att [A,C] = mlp_att.weight.
 * workspace: eamd_attloc_bwd_workspace(B, T, A, C) bytes of device memory, 16-byte aligned (per-workgroup partial sums).
 * accumulate != 0: d_enc_h and df are ADDED to (the caller keeps one running sum over the decoder steps of an utterance
 * batch - both are gradients of step-invariant tensors - instead of 101 fresh 33 MB tensors for autograd to add up).
 * Replaces the two N = C GEMMs of the step.  reference: autograd of rnn/attentions.py:329-365. */
int64_t eamd_attloc_bwd_workspace(int B, int T, int A, int C);
int eamd_attloc_bwd_energy_conv(const float* dctx, const float* dw_ext, const float* w, const float* enc_h, const float* th,
                                const float* gvec, float scaling, const float* conv, const float* w_att, float* de,
                                float* d_enc_h, float* df, float* dconv, float* dgvec, float* dgb, float* d_dec_proj,
                                float* dw_att, float* workspace, int accumulate, int B, int T, int A, int C, int E,
                                void* stream);
/* Dot-product attention (AttDot rnn/attentions.py:91-164, per head of AttMultiHeadDot :845-990):
 * e[b,t] = k[b,t,:] . q[b,:] on already tanh-activated k = tanh(mlp_k h), q = tanh(mlp_q z); -inf for t >= lens[b];
 * then w = softmax(scaling * e), ctx = sum_t w * v (eamd_att_ctx_*: the softmax / context half of eamd_attloc_*).
 * eamd_att_ctx_bwd: de [B,T], d_v [B,T,E] written, dsum[0] += sum de (= 0 analytically);
 * eamd_att_dot_energy_bwd: dk [B,T,A] written, dq [B,A] accumulated. */
int eamd_att_dot_energy_fwd(const float* k, const float* q, const int32_t* lens, float* e, int B, int T, int A,
                            void* stream);
int eamd_att_dot_energy_bwd(const float* de, const float* k, const float* q, float* dk, float* dq, int B, int T, int A,
                            void* stream);
int eamd_att_ctx_fwd(const float* e, const float* v, float scaling, float* w, float* ctx, int B, int T, int E,
                     void* stream);
int eamd_att_ctx_bwd(const float* dctx, const float* dw_ext, const float* w, const float* v, float scaling, float* de,
                     float* d_v, float* dsum, int B, int T, int E, void* stream);
/* backward stage 2, given dconv = df @ W_att [B,T,C]: d att_prev [B,R,T] and dconv_w [C,R,K] (accumulated). */
int eamd_attloc_bwd_conv(const float* dconv, const float* conv_w, const float* att_prev, float* d_prev, float* dconv_w,
                         int B, int T, int C, int K, int R, void* stream);
/* AttLocRec front end (attentions.py:690-696): pooled[b,c] = max_t relu(Conv2d(1,C,(1,K))(att_prev)[b,c,t]) and the
 * frame idx[b,c] of that maximum; backward: d_prev [B,T] and dconv_w [C,K] are ACCUMULATED (zero d_prev first). */
int eamd_attloc_convmax_fwd(const float* att_prev, const float* conv_w, float* pooled, int32_t* idx, int B, int T, int C,
                            int K, void* stream);
int eamd_attloc_convmax_bwd(const float* dpool, const float* pooled, const int32_t* idx, const float* att_prev,
                            const float* conv_w, float* d_prev, float* dconv_w, int B, int T, int C, int K, void* stream);

/* ---- RNN-Transducer ----------------------------------------------------------------------------
 * Joint network pointwise part: out[b,t,u,:] = act(enc[b,t,:] + dec[b,u,:]) (fp32 and/or bf16 copy for
 * the lin_out GEMM).  reference: transducer/joint_network.py:34-48, rnn_decoder.py:155-162.
 * Backward: d_enc[b,t,:] = sum_u dh * act'(pre), d_dec[b,u,:] = sum_t dh * act'(pre) (pre recomputed). */
int eamd_joint_fwd(const float* enc, const float* dec, float* out, void* out_bf16, int B, int T, int U, int J, int act,
                   void* stream);
int eamd_joint_bwd(const float* dh, const float* enc, const float* dec, float* d_enc, float* d_dec, int B, int T, int U,
                   int J, int act, void* stream);
/* Transducer loss on raw joint logits[B,T,U,V] (U = max label length + 1), log-softmax inside.
 * reference: transducer/loss.py:8-79 (warp-transducer RNNTLoss(blank) semantics), utils.py:9-53
 * (labels[B,U-1] int32 padded with blank, tlens[B] encoder lengths, ulens[B] label lengths).
 * loss[b] = -log P(y_b | x_b).  grad (optional, may alias logits) receives
 * scale * gscale_dev[0] * d loss[b] / d logits (gscale_dev may be NULL = 1).
 * workspace: eamd_rnnt_workspace(B,T,U) floats. */
int64_t eamd_rnnt_workspace(int B, int T, int U);
int eamd_rnnt_loss(const float* logits, const int32_t* labels, const int32_t* tlens, const int32_t* ulens,
                   float* workspace, float* loss, float* grad, int B, int T, int U, int V, int blank,
                   const float* gscale_dev, float scale, void* stream);
/* the gradient pass alone, reusing the workspace an eamd_rnnt_loss call (grad = NULL) filled for the same logits */
int eamd_rnnt_grad(const float* logits, const int32_t* labels, const int32_t* tlens, const int32_t* ulens,
                   const float* workspace, float* grad, int B, int T, int U, int V, int blank, const float* gscale_dev,
                   float scale, void* stream);
/* The same loss without the [B,T,U,V] tensor: the caller streams the joint logits through a buffer of `nrows`
 * consecutive lattice nodes (node = (b*T + t)*U + u) at a time - lin_out(act(lin_enc(h_enc) + lin_dec(h_dec))) of those
 * rows, recomputed in backward - and the three entry points below fill / consume the 5*B*T*U-float lattice workspace.
 * reference: transducer/rnn_decoder.py:160-165 + transducer/loss.py:74-76 (which materialise the logits). */
int eamd_rnnt_node_stats(const float* logits_rows, const int32_t* labels, float* workspace, int64_t node0, int64_t nrows,
                         int B, int T, int U, int V, int blank, void* stream);
int eamd_rnnt_alpha_beta(float* workspace, const int32_t* tlens, const int32_t* ulens, float* loss, int B, int T, int U,
                         void* stream);
/* eamd_rnnt_node_stats for logits that were never stored: part / zlab / zblank are what eamd_gemm's row-statistics
 * epilogue (epilogue 7, eamd_gemm_t.stats with col[m] = the node's next label or -1, fix = blank) left for the nrows nodes
 * node0 .. node0 + nrows - 1; tiles_n = ceil(V / tile) of that launch. */
int eamd_rnnt_node_stats_part(const float* part, const float* zlab, const float* zblank, float* workspace, int64_t node0,
                              int64_t nrows, int tiles_n, int B, int T, int U, void* stream);
/* per-node coefficients (tot, gb, gl) of the lattice rows node0 .. node0 + nrows - 1 for eamd_gemm's epilogue 8, from the
 * workspace eamd_rnnt_alpha_beta completed: rowc [nrows][3] (tot = -inf marks a node outside the lattice or unreachable) and
 * col [nrows] = the node's next label or -1. */
int eamd_rnnt_row_coef(const int32_t* labels, const int32_t* tlens, const int32_t* ulens, const float* workspace, float* rowc,
                       int32_t* col, int64_t node0, int64_t nrows, int B, int T, int U, void* stream);
int eamd_rnnt_node_grad(const float* logits_rows, float* grad_rows, void* grad_rows_bf16, const int32_t* labels,
                        const int32_t* tlens, const int32_t* ulens, const float* workspace, int64_t node0, int64_t nrows,
                        int B, int T, int U, int V, int blank, const float* gscale_dev, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer on flat fp32 arenas.  reference: transformer/optimizer.py:12-75 (NoamOpt),
 * espnet2/schedulers/warmup_lr.py:10-53, trainer.py:430-467 (clip + non-finite skip).
 * state: 8 device floats [step, lr, bias_corr1, bias_corr2, grad_norm, skipped, clip_coef, -].
 * ------------------------------------------------------------------------------------------ */
int eamd_grad_norm(const float* g, int64_t n, float* workspace, float* gnorm_out, void* stream);
int eamd_sched_step(float* state, const float* gnorm, int mode, float base_lr, float factor, float dmodel,
                    float warmup, float beta1, float beta2, float max_norm, void* stream);
/* p_bf16 (optional): bf16 shadow of the updated parameters, written by the same pass */
int eamd_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const float* state,
                   float beta1, float beta2, float eps, float weight_decay, void* stream);
/* torch.optim.Adadelta step on flat arenas (the RNN recipes' optimizer: espnet/asr/pytorch_backend/asr.py:505-508).
 * lr = state[1], clip coefficient = state[6] (eamd_sched_step), eps = state[7] (written by the host; the trainer's
 * eps decay, asr.py:798-830, rescales it in place); skipped when the gradient norm state[4] is not finite. */
int eamd_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta, void* p_bf16, int64_t n,
                       const float* state, float rho, float weight_decay, void* stream);
/* g[i] += sigma * N(0,1) over the flat gradient arena (espnet2/torch_utils/add_gradient_noise.py:4-31; the caller
 * computes sigma = eta / (iteration // duration + 1)^scale_factor).  Counter-based draws keyed by the device step
 * counter and `salt`: no generator state, graph-replayable. */
int eamd_add_gradient_noise(float* g, int64_t n, float sigma, const uint64_t* step_dev, uint64_t salt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Integer / layout helpers (bit-exact).
 * ------------------------------------------------------------------------------------------ */
/* reference: transformer/add_sos_eos.py:12-31.  ys_in/ys_out are [B, L+1]; olen[b] = #labels. */
int eamd_add_sos_eos(const int64_t* ys_pad, int64_t* ys_in, int64_t* ys_out, int32_t* olen, int B, int L, int sos,
                     int eos, int ignore_id, void* stream);
/* reference: e2e_asr_transformer.py:274-284 (greedy CTC: groupby + drop blank). out padded with -1. */
int eamd_ctc_collapse(const int32_t* ids, const int32_t* hlens, int32_t* out, int32_t* outlen, int B, int T,
                      int blank, void* stream);
/* Conv2d(C,C,3,2) weight [Co][Ci][3][3] -> wf [9][Ci][Co] (tap = kh*3+kw) and wd [9][Co][Ci] with
 * taps in stride-parity class order (0,0)(0,2)(2,0)(2,2)|(0,1)(2,1)|(1,0)(1,2)|(1,1).
 * reference: transformer/subsampling.py:31 (second Conv2d). */
int eamd_conv2_weight_prep(const float* w, void* wf, void* wd, int Co, int Ci, int out_bf16, void* stream);
/* dw[Co][Ci][3][3] += dwf[9][Ci][Co] */
int eamd_conv2_weight_grad(const float* dwf, float* dw, int Co, int Ci, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESPNET_AMD_H_ */
