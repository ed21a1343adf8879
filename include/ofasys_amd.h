/*
 * ofasys_amd.h -- C ABI of libofasys_amd.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * OFASys unified encoder-decoder hot path.
 *
 * What this boundary replaces.  The reference has no C/FFI boundary of its own for this path except the
 * pybind11 fused-softmax extensions; everything else goes torch -> ATen -> cuBLAS/cuDNN.  Each entry point
 * below names the reference call site (relative to /root/reference/ofasys) whose arithmetic it carries.
 *
 * Conventions (carried over from the reference's extensions, module/fused_kernels/scaled_masked_softmax.h:476,
 * scaled_softmax_cuda.cu:84-99, scaled_masked_softmax.cpp:45-49):
 *   - the caller owns every buffer; the library never allocates device memory;
 *   - raw device pointers + explicit sizes / leading dimensions (in ELEMENTS), row-major;
 *   - work is enqueued on the caller's stream (`stream` is a hipStream_t passed as void*), no hidden sync,
 *     no internal threads;
 *   - every function returns an int status: 0 ok, nonzero = error (the Python wrapper raises RuntimeError,
 *     mirroring TORCH_CHECK); shape preconditions are status codes, not asserts;
 *   - dtype is an ofa_dtype; accumulation is always fp32.
 */
#ifndef OFASYS_AMD_H
#define OFASYS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element types.  Every entry point takes fp32, bf16 or fp16 storage (`dtype`), accumulates in fp32 and -- for the 16-bit types --
 * multiplies on the matrix cores (v_mfma_f32_32x32x16_bf16 / _f16).  fp16 is the reference trainer's default precision
 * (config/default_trainer.yaml:7-25) and the only dtype it routes to its fused-softmax extensions (multihead_attention.py:83-91);
 * the fused attention kernels (ofa_attn_fwd / _bwd / _bwd_prep) are 16-bit only.  "bf16" in a comment below means "the 16-bit
 * dtype of the call" unless it says otherwise. */
typedef enum { OFA_F32 = 0, OFA_BF16 = 1, OFA_F16 = 2 } ofa_dtype;

enum {
  OFA_OK = 0,
  OFA_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, misaligned leading dimension) */
  OFA_ERR_UNSUPPORTED = 2, /* valid request this build has no kernel for (e.g. head_dim != 64 in the fused path) */
  OFA_ERR_LAUNCH = 3       /* hipLaunch / runtime error; see ofa_last_error() */
};

/* GEMM epilogue flags (ofa_gemm.flags) */
enum {
  OFA_GEMM_BIAS_COL = 1,   /* C[m][n] += bias[n]  (nn.Linear bias)                              */
  OFA_GEMM_BIAS_ROW = 2,   /* C[m][n] += bias[m]  (transposed-output projections)                */
  OFA_GEMM_ACCUM = 4,      /* C = result + C_old  (gradient accumulation)                        */
  OFA_GEMM_FORCE_SIMPLE = 8, /* use the exact-fp32-FMA VALU kernel even for bf16 (tests)         */
  OFA_GEMM_OUT_F32 = 16,   /* bf16 inputs, fp32 output                                           */
  OFA_GEMM_A_KPAD_ZERO = 32, /* caller guarantees A[m][K..roundup8(K)) == 0 (padded logits-gradient rows) */
  OFA_GEMM_NO_LDS_DMA = 64, /* force the register-staged loop (tests / A-B measurements)              */
  OFA_GEMM_DEFER_REDUCE = 128 /* split-K: leave the fp32 partials in ws, the caller folds them (ofa_fold_batched) */
};

int ofa_version(void);
const char* ofa_last_error(void);

/* ---- LayerNorm: torch.nn.LayerNorm(eps, affine) -- module/layer_norm.py:27-32; saves mean and rstd in fp32
 * exactly as the (never built) apex kernel does, fused_kernels/layer_norm_cuda.cpp:136-138. cols <= 8192. */
int ofa_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int cols, float eps, int dtype, void* stream);
/* dgamma/dbeta are [cols] in `dtype` (accumulate != 0: added to their current contents, i.e. straight into a gradient
 * arena); `ws` is fp32 scratch of ofa_layernorm_bwd_ws_rows()*cols floats.  cols <= 2048 (fp32) / 4096 (bf16) per
 * row-part: rows up to 4x that are split over four waves. */
int ofa_layernorm_bwd_ws_rows(void);
/* dres (optional, same shape as x): added to dx -- the gradient that reaches x through a residual branch taken before
 * the LayerNorm (pre-LN layers: `residual = x; x = LN(x)`, transformer_layer.py:159-161), saving the separate add. */
int ofa_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, void* dgamma, void* dbeta, float* ws, int64_t rows, int cols,
                      int accumulate, int dtype, void* stream);
/* y = LayerNorm(gelu(h)) -- transformer_layer.py:194-197 (fc1 -> GELU (module/gelu.py:18-19, fp32 erf) -> ffn_layernorm). */
int ofa_gelu_layernorm_fwd(const void* h, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                           int64_t rows, int cols, float eps, int dtype, void* stream);
/* dbias (optional, [cols]): column sums of dh = gradient of the bias of the Linear that produced h (fc1), same
 * accumulate rule as dgamma/dbeta. */
int ofa_gelu_layernorm_bwd(const void* dy, const void* h, const void* gamma, const float* mean, const float* rstd,
                           void* dh, void* dgamma, void* dbeta, void* dbias, float* ws, int64_t rows, int cols,
                           int accumulate, int dtype, void* stream);

/* ---- residual join of a pre-LN sub-block (transformer_layer.py:167-208, 438-494), one pass each way:
 *   y = residual + dropout_p(LN_a(x)),  z = LN_b(y);   LN_a (gamma_a/beta_a) and LN_b (gamma_b/beta_b, z) are optional.
 * stats: fp32 [4][rows] = mean_a, rstd_a, mean_b, rstd_b (kept for the backward).  Dropout as ofa_dropout_add_fwd
 * (Philox position offset + *offset_base).  Rounding points match the unfused kernels (y is bit-identical without LN_a; with it, and for
 * every gradient, the row sums are grouped differently: fp32 rounding noise).
 * Backward: dy / dz = gradients of y / z (NULL = zero); dres = gradient of residual (== total gradient of y), dx = gradient
 * of x; ws: fp32 [5][ofa_join_bwd_slots()][cols] partial rows of dgamma_a, dbeta_a, dgamma_b, dbeta_b and (want_dx_colsum) the
 * column sums of dx -- the bias gradient of the Linear that produced x -- for ofa_fold_batched.
 * residual == NULL (and dres == NULL in backward): y = dropout_p(LN_a(x)) -- the adaptor post-hook's `dropout(layernorm_embedding(embed))`
 * (adaptor/base.py:178-182) in one pass instead of a LayerNorm and a dropout kernel.
 * keep_bits (optional, ofa_join_keep_bytes() bytes; 0 bytes: the shape has no such buffer, pass NULL): the forward leaves the dropout
 * decisions there, one bit per element in the kernels' own lane order, and a backward given the same buffer reads them instead of running the
 * Philox generator again (which is what bounds these kernels, not HBM); NULL on either side: the mask is regenerated, same bits. */
int64_t ofa_join_keep_bytes(int64_t rows, int cols, int dtype);
int ofa_join_fwd(const void* x, const void* residual, const void* gamma_a, const void* beta_a, const void* gamma_b,
                 const void* beta_b, void* y, void* z, float* stats, uint8_t* keep_bits, int64_t rows, int cols, float eps, float p,
                 uint64_t seed, uint64_t offset, const int64_t* offset_base, int dtype, void* stream);
int ofa_join_bwd_slots(int64_t rows, int cols, int dtype);
int ofa_join_bwd(const void* dy, const void* dz, const void* x, const void* y, const void* gamma_a, const void* gamma_b,
                 const float* stats, const uint8_t* keep_bits, void* dres, void* dx, float* ws, int64_t rows, int cols, float p,
                 uint64_t seed, uint64_t offset, const int64_t* offset_base, int want_dx_colsum, int dtype, void* stream);

/* ---- GEMM: C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+bias) (+C).  Replaces F.linear / torch.bmm / matmul:
 * multihead_attention.py:199-217,308,338,346; transformer_layer.py:194,202; adaptor/general.py:223-243.
 *   transA == 0: A is [M,K] row-major (lda >= K);  transA == 1: A is stored [K,M] (lda >= M).
 *   transB == 0: B is [K,N] row-major (ldb >= N);  transB == 1: B is stored [N,K] (ldb >= K)  <- nn.Linear weight.
 * batch > 1: strided-batched (strides in elements).  Two-level batching for per-(batch, head) products on [B,T,heads*hd]
 * rows without copies: batch index z addresses operand X at (z / batch_inner)*strideX2 + (z % batch_inner)*strideX
 * (batch_inner <= 0 or >= batch: single level).  bf16 runs on MFMA (v_mfma_f32_32x32x16_bf16) when every
 * leading dimension is a multiple of 8 elements; fp32 (and OFA_GEMM_FORCE_SIMPLE) run an exact fp32-FMA kernel.
 * `ws`/`ws_bytes`: optional fp32 scratch that enables split-K for skinny outputs (may be NULL). */
int ofa_gemm(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int transA, int transB,
             int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
             int batch_inner, int64_t strideA2, int64_t strideB2, int64_t strideC2,
             float alpha, int flags, int dtype, void* ws, int64_t ws_bytes, void* stream);

/* ofa_gemm (one product, no batch) that ALSO leaves the per-column (sum, sum of squares) of its rounded 16-bit output as partial rows
 * [groups][2][N] fp64, one per wave block of rows, written by the kernel's epilogue from the tile it holds anyway -- the statistics
 * pass of a BatchNorm behind a convolution (module/resnet.py:105-128) for free.  *groups (HOST int, set before the call returns) is
 * the number of partial rows written, or 0 when the selected plan cannot produce them (split-K, fp32 output, accumulation, N % 8,
 * more than max_groups rows): the product is computed either way and the caller then runs its own statistics pass. */
int ofa_gemm_colstat(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int transA, int transB,
                     int64_t lda, int64_t ldb, int64_t ldc, float alpha, int flags, int dtype, void* ws, int64_t ws_bytes,
                     double* partial, int max_groups, int* groups, void* stream);

/* ---- grouped weight-gradient products: up to 16 independent  slabs_p[s][m][n] = sum over K-slice s of a_p^T b_p  in ONE
 * launch (256 x 256 eight-wave tiles).  The nn.Linear weight gradients of one Transformer layer (dW = dY^T X, what autograd
 * computes for transformer_layer.py:194,202 and multihead_attention.py:199-217,346) are 9-36 such tiles each: launched alone
 * each has to be cut into 3-7 short K-slices to occupy the chip; together they fill it with ~2 long slices each.
 * The slabs are finished by ofa_fold_batched (out (+)= alpha * sum_s slabs[s]).  16-bit operands; any k >= 1 (a row count that
 * is not a multiple of the 64-row K tile is completed with zero rows inside the kernel), m, n, lda, ldb % 8 == 0, 16-byte aligned
 * pointers.  `items` is a HOST array. */
typedef struct ofa_gemm_group_item {
  const void* a;     /* [k, m] rows (lda >= m): output-gradient rows dY */
  const void* b;     /* [k, n] rows (ldb >= n): layer-input rows X */
  float* slabs;      /* [splits][m][n] fp32, written (not accumulated) */
  int64_t lda, ldb;
  int32_t m, n, k;
  int32_t splits;    /* filled by ofa_gemm_group_plan */
  void* out;         /* optional, with out_alpha / ldo: when the plan leaves this product ONE K-slice (splits == 1) the kernel adds */
  int64_t ldo;       /* out_alpha * a^T b straight onto out [m, n] (16-bit, row stride ldo, 16-byte aligned) in its epilogue -- no slab, */
  float out_alpha;   /* no fold launch (the weight gradients of a small micro-batch: every product is one slice).  NULL: always slabs */
} ofa_gemm_group_item;
/* host only: validates the items and fills `splits` (one K-slice length for the group: <= 256 workgroups in total) */
int ofa_gemm_group_plan(ofa_gemm_group_item* items, int n, int dtype);
int ofa_gemm_group_tn(const ofa_gemm_group_item* items, int n, int dtype, void* stream);

/* ---- the reference's fused softmax extensions, wave64 re-derivations.
 * x,y: [b, np, sq, sk]; softmax over sk of scale*x, fp32 accumulate.  sk <= 4096 (scaled_masked_softmax_cuda.cu:47-52). */
int ofa_scaled_softmax_fwd(const void* x, void* y, float scale, int b, int np, int sq, int sk, int dtype, void* stream);
/* dx = scale*(dy*y - y*sum(dy*y)); dx may alias dy (in place, scaled_softmax_cuda.cu:84-99). */
int ofa_scaled_softmax_bwd(const void* dy, const void* y, void* dx, float scale, int b, int np, int sq, int sk,
                           int dtype, void* stream);
/* mask: uint8 [mask_b (b or 1), 1, sq, sk]; masked scores are replaced by -10000.0 (scaled_masked_softmax.h:269-273). */
int ofa_scaled_masked_softmax_fwd(const void* x, const uint8_t* mask, void* y, float scale, int b, int np, int sq,
                                  int sk, int mask_b, int dtype, void* stream);
/* replaces scaled_masked_softmax.cpp:60-77 `backward(output_grads, softmax_results, scale)`: same arithmetic as
 * ofa_scaled_softmax_bwd (the mask does not enter the backward); dx may alias dy -- the reference's backward is completely
 * in place on output_grads (scaled_masked_softmax_cuda.cu:98-117). */
int ofa_scaled_masked_softmax_bwd(const void* dy, const void* y, void* dx, float scale, int b, int np, int sq, int sk,
                                  int dtype, void* stream);
/* x,y: [attn_batches, sq, sq]; implicit causal mask, masked outputs are zero (scaled_upper_triang_masked_softmax.h:113-230). */
int ofa_scaled_upper_triang_masked_softmax_fwd(const void* x, void* y, float scale, int attn_batches, int sq,
                                               int dtype, void* stream);
/* replaces scaled_upper_triang_masked_softmax.cpp:49-64 `backward`: dy,y,dx [attn_batches, sq, sq]; the strictly upper triangle
 * of dy is never read and dx is zero there (scaled_upper_triang_masked_softmax.h:232-329); dx may alias dy (in place,
 * scaled_upper_triang_masked_softmax_cuda.cu:68-95). */
int ofa_scaled_upper_triang_masked_softmax_bwd(const void* dy, const void* y, void* dx, float scale, int attn_batches,
                                               int sq, int dtype, void* stream);
int ofa_get_batch_per_block(int sq, int sk, int b, int np); /* scaled_masked_softmax.h:426-438, kept for API parity */

/* ---- attention-score softmax of the slow path, multihead_attention.py:311-334:
 * p = softmax_fp32( x*scale + bias + causal(-inf above diagonal) ; key_padding -> -inf ) over S.
 * x,bias,p: [BA, T, S] (bias may be NULL); kpm: uint8 [B, S] or NULL (BA = B*heads). */
int ofa_attn_softmax_fwd(const void* x, const void* bias, const uint8_t* kpm, void* p, float scale, int BA, int heads,
                         int T, int S, int causal, int dtype, void* stream);

/* ---- fused attention (bf16, head_dim 64): multihead_attention.py:218-346 without materialising [BA,T,S].
 * q: [B, T, heads*64] rows (ld = ldq elements); k, v: [B, S, heads*64] rows (both with ld = ldk);
 * bias: optional dense [B*heads, T, S] additive bias (same dtype); kpm: optional uint8 [B,S]; c_attn: optional
 * [heads] per-head output scale (:342-345), fp32 or bf16 as c_attn_dtype says (the parameter itself, no cast kernel). out: [B, T, heads*64] (ld = ldo); lse: fp32 [B*heads, Tpad].  scale
 * multiplies q.k (the reference pre-scales q, :218) and must be positive (the kernel takes row maxima on the raw scores; a
 * non-positive scale is refused).  Tpad: a multiple of 32 covering T.
 * Attention dropout is not supported here (the reference default is attention_dropout = 0.0).
 * Ragged ("packed rows") mode, seg != NULL: the reference pads every sample to the longest and masks the padding
 * (multihead_attention.py:319-326); here the batch may arrive packed instead.  seg: device int32 [B][4] = {q_off, q_len, k_off,
 * k_len} (16-byte aligned; q_off a multiple of 4).  q / out are [rows_q, ld] with sample b's queries in rows q_off .. q_off+q_len-1,
 * k / v are [rows_k, ld] likewise; lse is fp32 [heads, Tpad] indexed by the packed query row (Tpad >= rows_q); T, S are upper
 * bounds of q_len, k_len (they size the launch grid); rows_q / rows_k: total packed rows; bias and kpm must be NULL.  The rows of
 * `out` outside every segment (alignment / bucket filler) are written as zeros. */
int ofa_attn_fwd(const void* q, const void* k, const void* v, const void* bias, const uint8_t* kpm,
                 const void* c_attn, int c_attn_dtype, void* out, float* lse, int B, int heads, int T, int S, int Tpad,
                 int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q, int rows_k,
                 int dtype, void* stream);
/* Backward.  lse: fp32 [B*heads, Tpad] as written by ofa_attn_fwd (base-2 log-sum-exp of the scaled, biased, masked
 * scores); delta: fp32 [B*heads, Tpad] = rowsum(dO*O); dout: [B,T,heads*64] rows (ld = ldo).
 * out != NULL (the forward output O, rows like dout): the dQ kernel computes delta from dO and O on its way in and WRITES it
 * (rows t < T; in ragged mode also zeros in the filler rows) -- for the dK/dV kernel and ofa_c_attn_grad; no separate pass.
 * out == NULL: delta is an input, filled by ofa_attn_bwd_prep beforehand.
 * Writes dq [B,T,D] (ld = ldq), dk, dv [B,S,D] (ld = ldk); dbias (optional, [B*heads,T,S]) receives dS.
 * No transposed operand copies are needed: the kernels transpose tiles on the LDS read (ds_read_b64_tr_b16).
 * seg != NULL: ragged mode as in ofa_attn_fwd (delta from ofa_attn_bwd_prep(B = 1, T = rows_q): [heads, Tpad] by packed row;
 * dq / dk / dv rows outside every segment are written as zeros). */
int ofa_attn_bwd_prep(const void* dout, const void* out, float* delta, int B, int heads, int T, int Tpad, int64_t ldo,
                      int dtype, void* stream);
int ofa_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const void* bias, const uint8_t* kpm,
                 const void* c_attn, int c_attn_dtype, const float* lse, float* delta, const void* out, void* dq, void* dk,
                 void* dv, void* dbias, int B, int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo,
                 float scale, int causal, const int32_t* seg, int rows_q, int rows_k, int dtype, void* stream);

/* ofa_attn_bwd that also leaves the COLUMN SUMS of its outputs as fp32 partial rows for ofa_fold_batched -- the bias gradients of the
 * q / k / v projections (multihead_attention.py:199-217: q_proj / k_proj / v_proj carry biases) and the gradient of c_attn (:342-345) --
 * instead of a column-sum pass over dq / dk / dv and ofa_c_attn_grad afterwards.  One partial row per (sample, 128-row tile, wave):
 * cs_q: [ofa_attn_cs_slots(B, T)][cs_ldq], columns [64 h, 64 h + 64) of a row written by one wave of the dQ workgroup of head h;
 * cs_k, cs_v: [ofa_attn_cs_slots(B, S)][cs_ldk] likewise by the dK/dV kernel; cs_c: [ofa_attn_cs_slots(B, T)][heads], sum over the
 * wave's 32 rows of delta / c_attn[h].  Any of them may be NULL.  Every partial row is written on every call (empty tiles of a ragged batch:
 * zeros), the sums are over the fp32 values BEFORE their rounding to `dtype`, in a fixed order.  Ragged mode: B = the segment count, T / S = max_q / max_k. */
int ofa_attn_cs_slots(int B, int rows);
int ofa_attn_bwd_cs(const void* q, const void* k, const void* v, const void* dout, const void* bias, const uint8_t* kpm,
                    const void* c_attn, int c_attn_dtype, const float* lse, float* delta, const void* out, void* dq, void* dk,
                    void* dv, void* dbias, int B, int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo,
                    float scale, int causal, const int32_t* seg, int rows_q, int rows_k, int dtype, float* cs_q, int64_t cs_ldq,
                    float* cs_k, float* cs_v, int64_t cs_ldk, float* cs_c, void* stream);

/* ---- Attention with a batch-SHARED position bias.
 * The reference adds a dense bias [B*A, T, S] to the scores of every attention of its default configuration (use_self_attn_bias,
 * model/ofa.py:110-113; multihead_attention.py:308-311): abs-pos (pos_q_linear(pos) * pos_scaling) pos_k_linear(pos)^T per head
 * (adaptor/general.py:223-243; cross attention: model/transformer.py:280-299) + table_l[bucket[i][j]] on every slot's diagonal block
 * (general.py:265-280; text.py:101-104, image_resnet.py:116-128, video_image_sequence.py:187-204).  Position embeddings do not depend
 * on the batch row (text: arange; image / video: the patch grid), so that tensor is B copies of ONE [A, T, S] matrix per layer -- which
 * is what these entry points take, indexed by (head, query position, key position) for every sample; in ragged mode by the position
 * INSIDE the sample (so a batch whose valid positions are a prefix of each padded row packs without touching the bias).  Tb >= T,
 * Sb >= S.  Everything else as ofa_attn_fwd (seg != NULL: kpm must be NULL).
 *
 * The kernels do not read the row-major [heads, Tb, Sb] tensor but its two TILE-SWIZZLED images written by ofa_bias_build (which
 * also does the per-layer assembly of general.py:265-280): per head and 32 x 32 block, the 16 values MFMA lane l = (hi << 5 | i)
 * holds -- element r: row image  bias[32 qt + i][32 kt + (r & 3) + 8 (r >> 2) + 4 hi]  at ((h * nqt + qt) * nkt + kt) * 1024 + 16 l + r,
 * column image  bias[32 qt + (r & 3) + 8 (r >> 2) + 4 hi][32 kt + i]  at ((h * nkt + kt) * nqt + qt) * 1024 + 16 l + r,
 * nqt = ceil(Tb / 32), nkt = ceil(Sb / 32), positions outside [Tb, Sb] zero.  A wave then fetches a block's bias as 2 KiB of
 * consecutive bytes and feeds it to the score MFMAs as their initial accumulator. */
int ofa_attn_sbias_fwd(const void* q, const void* k, const void* v, const void* bias_swz_row, int Tb, int Sb, const uint8_t* kpm,
                       const void* c_attn, int c_attn_dtype, void* out, float* lse, int B, int heads, int T, int S, int Tpad,
                       int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q, int rows_k,
                       int dtype, void* stream);
/* Backward (out != NULL always: delta is computed on the way in and written).  dbias_sum (optional): [heads, Tb, Sb] in dbias_dtype
 * (OFA_F32, or a 16-bit code: rounded once, from the fp32 sum) = sum over the batch of dS -- the gradient of the shared bias.  The reference obtains it by materialising dS as [B*A, T, S] and
 * reducing the expand; here a third kernel walks (a chunk of) the batch per [128 x 64] tile of one head, recomputes S and dP of that
 * tile (lse / delta are known) and accumulates dS in registers: no [B*A,T,S] tensor, no atomics, bitwise reproducible.
 * bias: the row-major [heads, Tb, Sb] tensor in `dtype` (read once per tile by that third kernel; may be NULL when dbias_sum is).
 * ws / ws_bytes: see ofa_attn_sbias_chunks. */
int ofa_attn_sbias_bwd(const void* q, const void* k, const void* v, const void* dout, const void* bias, const void* bias_swz_row,
                       const void* bias_swz_col, int Tb, int Sb, const uint8_t* kpm, const void* c_attn, int c_attn_dtype,
                       const float* lse, float* delta, const void* out, void* dq, void* dk, void* dv, void* dbias_sum, int dbias_dtype,
                       float* ws, int64_t ws_bytes, int B, int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                       int causal, const int32_t* seg, int rows_q, int rows_k, int dtype, void* stream);
/* ... with the column-sum partial rows of ofa_attn_bwd_cs */
int ofa_attn_sbias_bwd_cs(const void* q, const void* k, const void* v, const void* dout, const void* bias, const void* bias_swz_row,
                          const void* bias_swz_col, int Tb, int Sb, const uint8_t* kpm, const void* c_attn, int c_attn_dtype,
                          const float* lse, float* delta, const void* out, void* dq, void* dk, void* dv, void* dbias_sum, int dbias_dtype,
                          float* ws, int64_t ws_bytes, int B, int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                          int causal, const int32_t* seg, int rows_q, int rows_k, int dtype, float* cs_q, int64_t cs_ldq, float* cs_k,
                          float* cs_v, int64_t cs_ldk, float* cs_c, void* stream);
/* Batch chunks the dS-sum kernel of ofa_attn_sbias_bwd cuts B samples into (short sequences: the [128 x 64] tiles of the heads alone
 * would leave the chip idle).  1 and dbias_dtype == OFA_F32 or == dtype: it writes dbias_sum itself, ws may be NULL.  Otherwise ws must hold
 * n * heads * Tb * Sb floats (the chunks' partial sums, folded in chunk order -- and cast -- by ofa_fold_batched inside the call). */
int ofa_attn_sbias_chunks(int B, int heads, int Tb, int Sb);
/* Gradient of the per-head scale c_attn (multihead_attention.py:58, 342-345: attn[t,b,h,:] *= c_attn[h]; O = c * PV, so
 * d c[h] = sum_{b,t} rowsum(dO*O)[b,h,t] / c[h]) from the delta rows of
 * ofa_attn_bwd_prep: dc[h] (+)= sum_b sum_{t<T} delta[(b*heads+h)*ld + t] / c_attn[h]; dc has c_attn's dtype. */
int ofa_c_attn_grad(const float* delta, const void* c_attn, void* dc, int B, int heads, int T, int64_t ld, int accumulate,
                    int c_attn_dtype, void* stream);
/* Decode-time attention (incremental branch, multihead_attention.py:241-279 + 308-346): ONE query row per (batch, head)
 * against a key/value cache.  q, out: [B, heads*64]; k, v: rows of a [B, capacity, ld] cache -- element (b, s, c) at
 * b*k_batch_stride + s*ldk + c -- of which the first S rows are valid; bias: optional [B*heads, S] additive row (the last
 * row of the relative-position bias); kpm: optional uint8 [B, kpm_ld] (non-zero = padded key); c_attn: optional per-head
 * scale (fp32 or bf16 per c_attn_dtype); probs: optional [B*heads, S] softmax output (need_weights).  Softmax in fp32;
 * fp32 and bf16 tensors; head_dim must be 64. */
int ofa_attn_decode(const void* q, const void* k, const void* v, const void* bias, const uint8_t* kpm, const void* c_attn,
                    int c_attn_dtype, void* out, void* probs, int B, int heads, int head_dim, int S, int64_t ldk,
                    int64_t k_batch_stride, int64_t kpm_ld, float scale, int dtype, void* stream);
/* out[b][i] = mean over heads of p[b][a][i], i < n  (head-averaged attention weights, multihead_attention.py:347-351). */
int ofa_mean_heads(const void* p, void* out, int B, int heads, int64_t n, int dtype, void* stream);
/* x: [B, T, C] rows (ld elements) -> xt: [B, C, Tpad] (zero-filled for t >= T). */
int ofa_transpose_heads(const void* x, void* xt, int B, int T, int C, int Tpad, int64_t ld, int dtype, void* stream);

/* ---- embeddings: F.embedding gather (adaptor/text.py:119-127) and its dense-gradient scatter-add, deterministic
 * (a wave per (vocabulary row, id-list slice) scans the ids with ballots; no atomics). ids: int64.
 * present_ws: optional V bytes of scratch (rows that do not occur are skipped); slice_ws: optional fp32 scratch of
 * ofa_embedding_bwd_slices(V, D) * V * D floats -- small tables with thousands of hits per row are then reduced in slices. */
/* Row gather with zero fill: out[r, :] = index[r] >= 0 ? src[index[r], :] : 0 (index int64 [n]; src [src_rows, D]).  Packs the
 * non-pad rows of a padded batch (ofasys_amd/packing.py; the reference computes the padding, preprocessor/utils.py:75-113) and,
 * with the inverse index, scatters gradients back. */
int ofa_gather_rows(const void* src, const int64_t* index, void* out, int64_t n, int D, int64_t src_rows, int dtype,
                    void* stream);
/* The same gather over the VIRTUAL concatenation along the sequence of nparts <= 8 slot outputs (adaptor/general.py:245-282
 * `torch.cat(tuple(x.embed ...), dim=1)`): index[r] = b * T + t addresses the concatenated [batch, T, D], T = sum lens; part k is its own
 * contiguous [batch, lens[k], D] (srcs / lens: HOST arrays).  The concatenated tensor is never built.  ofa_scatter_rows_part is the
 * backward for one part: out [batch, nk, D] row (b, j) = src[inverse[b * T + start + j]] or 0 (src: gradient of the packed rows). */
int ofa_gather_rows_parts(const void* const* srcs, const int* lens, int nparts, const int64_t* index, void* out, int64_t n, int D,
                          int64_t batch, int dtype, void* stream);
int ofa_scatter_rows_part(const void* src, const int64_t* inverse, void* out, int64_t batch, int nk, int Ttot, int start, int D,
                          int64_t src_rows, int dtype, void* stream);
/* out[r, :] = weight[ids[r], :]  (F.embedding, adaptor/text.py:124-125); is_pad (optional, n bytes): is_pad[r] = ids[r] == pad_id -- the
 * padding mask the text adaptor derives from the same ids (adaptor/text.py:108-111), written by the same pass. */
int ofa_embedding_fwd(const void* weight, const int64_t* ids, void* out, int64_t n, int D, int64_t V, uint8_t* is_pad, int64_t pad_id,
                      int dtype, void* stream);
/* dweight[seg_row[s], :] (+)= sum over p in [seg_off[s], seg_off[s+1]) of dout[order[p], :]   (D <= 64 columns).
 * The embedding gradient of a lookup whose ids are the same every step -- the rel-pos bias `table[bucket[i][j]]` (adaptor/text.py:
 * 101-104, image_resnet.py:116-128): order = positions sorted by id (stable), one segment per distinct id, built once per lookup
 * by the caller.  One wave per segment, fixed order: deterministic; accumulate != 0 adds to dweight (the gradient arena). */
int ofa_segment_rowsum(const void* dout, const int32_t* order, const int32_t* seg_off, const int32_t* seg_row, void* dweight, int nseg,
                       int D, int accumulate, int dtype, void* stream);
int ofa_embedding_bwd_slices(int64_t V, int D);
int ofa_embedding_bwd(const void* dout, const int64_t* ids, void* dweight, int64_t n, int D, int64_t V,
                      int64_t padding_idx, uint8_t* present_ws, float* slice_ws, int dtype, void* stream);

/* ---- elementwise pieces of the layer (transformer_layer.py:167-208): */
int ofa_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);                  /* module/gelu.py:18-19 */
int ofa_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream);
/* y = residual + dropout_p(x); mask is regenerated from (seed, offset) with Philox4x32-10, nothing is stored.
 * offset_base (optional, device int64[1]) is added to `offset` on the device: the stream position of a captured
 * (hipGraph) train step lives there and is advanced by the step itself, so replays draw fresh masks. */
int ofa_dropout_add_fwd(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed,
                        uint64_t offset, const int64_t* offset_base, int dtype, void* stream);
int ofa_dropout_bwd(const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t offset,
                    const int64_t* offset_base, int dtype, void* stream);
/* y[r][c] = (a[r][c] + (b? b[rb][c]:0) + (vec? vec[c]:0)) * (rowmask && rowmask[r] ? 0 : 1)  -- adaptor/base.py:168-173,
 * model/transformer.py:110-112.  rb = r, or r % b_period when b_period > 0: b then holds ONE sample's rows (position embeddings are the
 * same for every sample of a batch: adaptor/text.py:124 `embed_positions(arange)`) and is never expanded to [B, T, D]. */
int ofa_add_rowvec_mask(const void* a, const void* b, const void* vec, const uint8_t* rowmask, void* y, int64_t rows,
                        int cols, int64_t b_period, int dtype, void* stream);
/* out[i] (+)= sum_{b < batch} x[b * n + i]  (fp32 accumulation, one rounding): the gradient of such a batch-shared tensor. */
int ofa_batch_sum(const void* x, void* out, int batch, int64_t n, int accumulate, int dtype, void* stream);

/* out[c] (out_dtype) (+)= alpha * sum_r x[r][c]  -- bias gradients of nn.Linear (autograd of multihead_attention.py:199-217). */
int ofa_colsum_ws_floats(int cols);
int ofa_colsum(const void* x, void* out, float* ws, int64_t rows, int cols, int64_t ld, float alpha, int accumulate,
               int dtype, int out_dtype, void* stream);

/* out[i] = sum_{k < n} inputs[k][i] (n <= 16 tensors of `numel` elements; `inputs` is a HOST array of device pointers; fp32
 * accumulation, one rounding): the gradient of a tensor several consumers read -- the abs-position bias that every layer of a stack
 * builds its attention bias from (adaptor/general.py:265-270, model/transformer.py:280-299) -- which autograd would sum with n - 1
 * pairwise adds. */
int ofa_add_n(const void* const* inputs, int n, void* out, int64_t numel, int dtype, void* stream);
/* y = a * b, b either [rows,cols] or a [cols] row vector (c_attn head scale, multihead_attention.py:342-345). */
int ofa_mul(const void* a, const void* b, void* y, int64_t rows, int cols, int b_rowvec, int dtype, void* stream);
/* y[r, :] = x[r, :] * scale[r / group] (fp32 scale per group of `group` consecutive rows): DropPath (module/droppath.py:
 * 40-60) on batch-major rows -- one keep/(1-p) factor per sample. */
int ofa_scale_row_groups(const void* x, const float* scale, void* y, int64_t rows, int cols, int64_t group, int dtype,
                         void* stream);
/* out[0] = sum(x[0..n)) in one deterministic pass (loss = sum of per-row losses). */
int ofa_reduce_sum_f32(const float* x, float* out, int64_t n, void* stream);
/* out[h] = sum_b sum_{t<T} x[(b*heads+h)*ld + t]  (gradient of c_attn from the attention row sums). */
int ofa_head_sum_f32(const float* x, float* out, int B, int heads, int T, int ld, void* stream);

/* ---- attention-bias assembly (adaptor/general.py:265-280): bias [B,A,T,T] (in place) gets values [n,n,A] added on
 * the diagonal block [start, start+n)^2 of every (b,a); the gradient reduces that block over the batch. */
/* Per-layer assembly of the batch-shared position bias, general.py:265-280 on ONE [heads, Tb, Sb] matrix:
 *   out = abs_bias (NULL: zeros);  out[:, s:s+n, s:s+n] += values_k[i][j][h]  for each slot k (values_k: [n, n, heads], `dtype`)
 * written as the row-major tensor (out, optional) and as the two swizzled images the ofa_attn_sbias_* kernels read (swz_row,
 * swz_col: ofa_bias_swz_elems(heads, Tb, Sb) elements each, 16-byte aligned; layout: see ofa_attn_sbias_fwd).  16-bit dtypes only;
 * slot blocks need Tb == Sb. */
typedef struct ofa_bias_slots {
  const void* values[8];
  const void* values2[8];  /* NULL: a dense slot.  Else an OUTER slot (video_image_sequence.py:187-204: frame-level + patch-level
                            * rel-pos tables): value(i, j) = values[i / inner][j / inner] + values2[i % inner][j % inner], values
                            * [n / inner, n / inner, heads], values2 [inner, inner, heads] -- summed in `dtype` first, like the
                            * reference's broadcast add, without ever forming the [n, n, heads] tensor */
  int32_t start[8];
  int32_t n[8];
  int32_t inner[8];
  int32_t count;
} ofa_bias_slots;
int64_t ofa_bias_swz_elems(int heads, int Tb, int Sb);
int ofa_bias_build(const void* abs_bias, const ofa_bias_slots* slots, void* out, void* swz_row, void* swz_col, int heads, int Tb,
                   int Sb, int dtype, void* stream);
/* Gradient of an OUTER slot from the (batch-summed) bias gradient dbias [heads, T, T]: d_frames [F, F, heads] = sums over the patch
 * pairs, d_patches [P, P, heads] = sums over the frame pairs of the slot's diagonal block (fp32 accumulation, fixed order). */
int ofa_bias_outer_grad(const void* dbias, void* d_frames, void* d_patches, int heads, int T, int start, int F, int P, int dtype,
                        void* stream);
int ofa_bias_block_add(void* bias, const void* values, int B, int A, int T, int start, int n, int dtype, void* stream);
int ofa_bias_block_grad(const void* dbias, void* dvalues, int B, int A, int T, int start, int n, int dtype, void* stream);
/* A slot's OWN per-sample bias (an adaptor whose forward() fills AdaptorOutput.self_attn_bias, which the post-hook leaves alone:
 * adaptor/base.py:183-189; summed into the layer bias at adaptor/general.py:276): bias [B,A,T,T][:, :, s:s+n, s:s+n] += values
 * [B,A,n,n], and the gradient of `values` = that block of dbias, contiguous. */
int ofa_bias_block_add_batch(void* bias, const void* values, int B, int A, int T, int start, int n, int dtype, void* stream);
int ofa_bias_block_slice(const void* dbias, void* dvalues, int B, int A, int T, int start, int n, int dtype, void* stream);

/* ---- patch embedding (adaptor/image_patch_embed.py:59-73): im2col of non-overlapping p x p patches.
 * img [B,C,H,W] -> col [B*(lead + (H/p)*(W/p)), Kpad] with K = C*p*p in (c,ph,pw) order (= Conv2d weight.view(D,-1)), zero pad; every
 * sample's rows start with `lead` all-zero rows (1: the class-token position, so that the projection and its weight gradient run over
 * the [B, 1 + N, D] rows the adaptor returns and `torch.cat((cls_token, x))` (:71-73) never happens). */
int ofa_im2col_patch(const void* img, void* col, int B, int C, int H, int W, int p, int Kpad, int lead, int dtype, void* stream);

/* ---- criterion (engine/criterion/cross_entropy.py:27-67): fp32 log-softmax + NLL(sum, ignore_index) per row.
 * logits [rows, V] (ld elements); writes lse[rows] and row_loss[rows] (0 for ignored rows). */
int ofa_cross_entropy_fwd(const void* logits, const int64_t* target, float* lse, float* row_loss, int64_t rows,
                          int64_t V, int64_t ld, int64_t ignore_index, int dtype, void* stream);
/* Both in ONE pass over the logits (one read, one write instead of two reads and a write): lse, row_loss AND
 * dlogits = (softmax - onehot) * grad_scale[0] (grad_scale: device scalar known before the forward -- the backward seed or the loss
 * scale; NULL = 1; zero for ignored rows and for columns V..ld-1).  16-bit logits of at most 65536 padded columns
 * (ofa_cross_entropy_fwd_grad_ok != 0), the row is held in the registers of a 1024-thread block. */
int ofa_cross_entropy_fwd_grad_ok(int64_t V, int64_t ld, int dtype);
int ofa_cross_entropy_fwd_grad(const void* logits, const int64_t* target, const float* grad_scale, float* lse, float* row_loss,
                               void* dlogits, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, int dtype, void* stream);
/* dlogits = (softmax - onehot) * grad_scale[0] (device scalar), zero for ignored rows and for columns V..ld-1. */
int ofa_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* grad_scale,
                          void* dlogits, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, int dtype,
                          void* stream);

/* label-smoothed cross entropy fused with the log-softmax (engine/criterion/label_smoothed_cross_entropy.py:62-191):
 * row_loss = (1-eps-eps_i)*nll + eps_i*smooth over the allowed vocabulary: all of it (cstart < 0), or [0,4) U
 * [cstart,cend) (constraint_range), intersected with the optional per-row byte mask cmask [rows, V].  row_cnt receives
 * the allowed count (needed by the backward); ignored rows give 0.  Backward: per-row weights row_w (optional; 0 drops a
 * row -- drop_worst) and the scalar grad_scale multiply the gradient. */
int ofa_ls_cross_entropy_fwd(const void* logits, const int64_t* target, float* lse, float* row_loss, float* row_nll,
                             float* row_cnt, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, float eps,
                             int64_t cstart, int64_t cend, const uint8_t* cmask, int dtype, void* stream);
int ofa_ls_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* row_cnt,
                             const float* row_w, const float* grad_scale, void* dlogits, int64_t rows, int64_t V, int64_t ld,
                             int64_t ignore_index, float eps, int64_t cstart, int64_t cend, const uint8_t* cmask, int dtype,
                             void* stream);
/* get_normalized_probs (model/ofa.py:287-299, module/utils.py:451-462): out fp32 [rows, V] = (log-)softmax_fp32(logits);
 * backward writes dlogits [rows, ld] in `dtype` (columns V..ld-1 zero). */
int ofa_probs_fwd(const void* logits, float* out, int64_t rows, int64_t V, int64_t ld, int log_probs, int dtype,
                  void* stream);
int ofa_probs_bwd(const float* dy, const float* y, void* dlogits, int64_t rows, int64_t V, int64_t ld, int log_probs,
                  int dtype, void* stream);

/* ---- train-step glue (engine/trainer.py:857-884, optim/adam.py:144-218, optim/fp16_optimizer.py): flat arenas. */
int ofa_sumsq_ws_floats(void);
int ofa_sumsq(const void* x, float* out /* fp32[1], accumulated into */, float* ws /* ofa_sumsq_ws_floats() floats */,
              int64_t n, int dtype, void* stream);
/* Scalar schedule of one update, on the device (one thread): gnorm = sqrt(gsq)/sample_size; sched[0] = (1/sample_size) *
 * min(1, clip_norm/(gnorm + 1e-6)) (trainer.py:857-884; no clipping when clip_norm <= 0); step += 1; sched[1] =
 * lr*sqrt(1-b2^step)/(1-b1^step); sched[2] = lr (adam.py:205-207).  sched (fp32[5]) feeds ofa_adam_step(step = 0).
 * Guard (trainer.py:866-876 raises FloatingPointError and skips optimizer.step): a non-finite gnorm or sample_size <= 0 gives
 * sched = [0, 0, lr, skip = 1], leaves `step` unchanged and adds 1 to sched[4] (count of skipped updates, for the host to poll);
 * otherwise sched[3] = 0. */
int ofa_step_schedule(const float* gsq, const double* sample_size, double* step, const double* lr, float* sched,
                      float* gnorm, float clip_norm, double beta1, double beta2, void* stream);
/* stats [3] fp64 += [count of target != pad, loss[0], the same count]: one micro-batch's share of the update's logging sums
 * [sample_size, loss_sum, ntokens] (engine/trainer.py:842-860, criterion/cross_entropy.py:55-67 with sentence_avg = False); loss: a
 * device fp32 scalar; target: int64 [n].  One launch, nothing on the host. */
int ofa_step_stats_add(double* stats, const float* loss, const int64_t* target, int64_t n, int64_t pad, void* stream);
/* The same with a DYNAMIC LOSS SCALE (engine/optim/fp16_optimizer.py:170-204 clip_grad_norm / step, dynamic_loss_scaler.py:9-70):
 * the arena holds loss_scale * (sum of gradients); loss_scaler: device fp64[8] = [loss_scale, iter, last_overflow_iter,
 * last_rescale_iter, overflows_since_rescale, fatal, -, -].  sched[0] = 1 / (loss_scale * sample_size), times clip_norm / gnorm when
 * gnorm > clip_norm > 0 (the fp16 optimizer's form: no 1e-6, no clamp); a non-finite gnorm runs check_overflow (scale /=
 * scale_factor when overflows / iters since the last rescale >= tolerance, floor `threshold` (<= 0: none), `fatal` = 1 instead of
 * the reference's FloatingPointError at min_loss_scale) and skips the update; otherwise update(): scale *= scale_factor every
 * scale_window updates since the last overflow. */
int ofa_step_schedule_scaled(const float* gsq, const double* sample_size, double* step, const double* lr, float* sched,
                             float* gnorm, double* loss_scaler, float clip_norm, double beta1, double beta2, double scale_factor,
                             double scale_window, double tolerance, double threshold, double min_loss_scale, void* stream);
/* Adam on fp32 master weights with grads of `dtype`; coef[0] = grad multiplier (world/sample_size and clip folded
 * in by the caller on device), writes the `dtype` model copy.  Weight decay as adam.py:209-210 (p -= wd*lr*p).
 * step >= 1: bias correction from (lr, step) on the host.  step == 0: coef is device fp32[>=4] = [grad multiplier,
 * lr*sqrt(1-b2^t)/(1-b1^t), lr, skip] -- the schedule state stays on the device (captured train steps); skip != 0 (set by
 * ofa_step_schedule on a non-finite gradient norm) leaves master weights, both moments and the model copy untouched. */
int ofa_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, void* model_param,
                  const float* coef, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int step, int dtype, void* stream);

/* ---- convolution stack of the image_resnet / video / audio adaptors (module/resnet.py:22-261, module/subsample.py:11-63).
 * Activations are NHWC rows [B*H*W, C]; a convolution is ofa_im2col (taps ordered (kh, kw, c), row length Kpad >=
 * kh*kw*C, zero padded) + ofa_gemm against the weight viewed as [Cout, kh*kw*C]; its input gradient is ofa_gemm +
 * ofa_col2im (gather formulation, deterministic).  x_nchw: the source is the [B,C,H,W] image itself. */
int ofa_conv_out_size(int in, int k, int stride, int pad);
int ofa_im2col(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
               int x_nchw, int dtype, void* stream);
int ofa_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
               int dtype, void* stream);
/* torch.nn.BatchNorm2d over the rows of [rows, C] (module/resnet.py:105-128): y = [relu]((x-mean)*rstd*gamma + beta
 * [+ residual]).  use_running == 0: batch statistics (biased variance), running buffers (fp32, optional) updated with
 * `momentum` and the unbiased variance; use_running != 0: eval mode.  mean/rstd: fp32 [C] outputs kept for backward.
 * ws: ofa_batchnorm_ws_floats(C) floats.  Backward: g = dy*[y>0] when relu; dres (optional) receives g.  beta (backward, optional;
 * relu layers without a residual input only): the gate is recomputed as [(x-mean)*rstd*gamma + beta > 0] and y is not read. */
int ofa_batchnorm_ws_floats(int C);
int ofa_batchnorm_fwd(const void* x, const void* gamma, const void* beta, const void* residual, void* y, float* mean,
                      float* rstd, float* running_mean, float* running_var, float* ws, int64_t rows, int C, float eps,
                      float momentum, int use_running, int relu, int dtype, void* stream);
int ofa_batchnorm_bwd(const void* dy, const void* y, const void* x, const void* gamma, const float* mean, const float* rstd,
                      void* dx, void* dres, void* dgamma, void* dbeta, float* ws, int64_t rows, int C, int batch_stats,
                      int relu, int accumulate, const void* beta, int dtype, void* stream);
/* SyncBatchNorm (adaptor/image_resnet.py:53-56, 87-90 `sync_bn` -> module/layer.py:26-27 nn.SyncBatchNorm): the same arithmetic in
 * two phases each way, around the CALLER's all-reduce (SUM over the data-parallel ranks) of `sums`:
 *   forward   fwd_stats: sums [2*C + 1] fp64 = this rank's (sum x, sum x^2, row count)  -> all-reduce -> fwd_apply: mean / rstd /
 *             running buffers from the reduced sums (the row count is read on the device: no host sync, ranks may differ), y for
 *             this rank's `rows` rows;
 *   backward  bwd_stats: sums [2][C] fp32 = this rank's (sum g, sum g*xhat); dgamma / dbeta from them (rank-local, like every other
 *             gradient before the gradient exchange) -> all-reduce -> bwd_dx: dx (and dres) with the reduced sums / *total_rows
 *             (device pointer: element 2*C of the forward's reduced sums).
 * fwd_apply with groups > 0 is also the BatchNorm forward BEHIND A CONVOLUTION whose GEMM left the column statistics of its output
 * as `groups` partial rows [groups][2][C] fp64 (ofa_gemm_colstat): no statistics pass over the activation at all. */
int ofa_batchnorm_fwd_stats(const void* x, double* sums, float* ws, int64_t rows, int C, int dtype, void* stream);
int ofa_batchnorm_fwd_apply(const void* x, const void* gamma, const void* beta, const void* residual, void* y, float* mean,
                            float* rstd, float* running_mean, float* running_var, const double* sums, int groups, int64_t rows,
                            int C, float eps, float momentum, int relu, int dtype, void* stream);
int ofa_batchnorm_bwd_stats(const void* dy, const void* y, const void* x, const void* gamma, const float* mean, const float* rstd,
                            float* sums, void* dgamma, void* dbeta, float* ws, int64_t rows, int C, int relu, int accumulate,
                            const void* beta, int dtype, void* stream);
int ofa_batchnorm_bwd_dx(const void* dy, const void* y, const void* x, const void* gamma, const float* mean, const float* rstd,
                         const float* sums, void* dx, void* dres, int64_t rows, const double* total_rows, int C, int relu,
                         const void* beta, int dtype, void* stream);
/* MaxPool2d(K, stride, pad) on NHWC; arg: one byte per output element (arg-max tap), consumed by the backward. */
int ofa_maxpool_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int K, int stride, int pad, int dtype,
                    void* stream);
int ofa_maxpool_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                    int dtype, void* stream);
/* y = relu(x) (gate == NULL), or y = x*[gate > 0] (its backward with gate = the forward output). */
int ofa_relu(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream);

/* ---- batched fold of fp32 partial rows: out[c] (+)= alpha * sum_{s < nslots} part[s*stride + c], c < cols, for up to
 * any number of jobs in as few launches as possible (56 jobs per launch).  Producers that were asked to leave their
 * partials in place (ofa_layernorm_bwd / ofa_gelu_layernorm_bwd / ofa_colsum with accumulate == OFA_DEFER_FOLD,
 * ofa_gemm with OFA_GEMM_DEFER_REDUCE) are finished by this call; slot order is fixed, results are identical to the
 * immediate reduce.  `jobs` is a HOST array. */
typedef struct ofa_fold_job {
  const float* part;   /* device fp32 partial rows */
  void* out;           /* device output, fp32 or bf16 */
  int64_t cols;
  int64_t stride;      /* elements between consecutive partial rows (>= cols) */
  int32_t nslots;
  int32_t accumulate;  /* != 0: added to the current contents of out */
  float alpha;
  int32_t out_dtype;   /* OFA_F32 / OFA_BF16 */
} ofa_fold_job;
int ofa_fold_batched(const ofa_fold_job* jobs, int njobs, void* stream);
/* Batched device-to-device copy: dst_i[0 .. bytes_i) = src_i[0 .. bytes_i) for any number of (dense) buffers in as few launches as
 * possible (96 jobs per launch) -- the static inputs of a replayed step graph (engine/trainer.py's _prepare_sample moves a batch to
 * the device tensor by tensor; a replay has to copy each into the graph's input, ~3.5 us per launch).  `jobs` is a HOST array. */
typedef struct ofa_copy_job {
  const void* src;
  void* dst;
  int64_t bytes;
} ofa_copy_job;
int ofa_copy_batched(const ofa_copy_job* jobs, int njobs, void* stream);
#define OFA_DEFER_FOLD 2
/* partial-row count (`nslots`) the corresponding call writes; ws layouts: LayerNorm [q][nslots][cols] with q = dgamma,
 * dbeta(, dbias); colsum [nslots][cols]; split-K GEMM [splits][M][(N+3)&~3] (ofa_gemm_splits == 1: no partials). */
int ofa_layernorm_bwd_slots(int64_t rows, int cols, int dtype, int gelu);
int ofa_colsum_slots(int64_t rows);
int ofa_gemm_splits(int M, int N, int K, int transA, int transB, int batch, int flags, int dtype, int64_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif /* OFASYS_AMD_H */
