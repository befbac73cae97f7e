/* rat_hip.h — C ABI of librat_hip.so: the MI355X (gfx950) hot path of RAT_m2.
 *
 * The reference (YushenLi807/WWW24-RAT) is pure Python on torch ATen and has no FFI; the drop-in boundary
 * is its FuxiCTR model-plugin API (run_expid.py:75-76 -> fuxictr/pytorch/models/RAT_m2.py).  This header is
 * what a ctypes binding inside that plugin calls instead of the ATen op sequences cited per entry point.
 * All paths below are relative to /root/reference.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the caller (PyTorch's allocator)
 *    owns every buffer including workspaces; the library allocates and frees nothing;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises the
 *    device, is re-entrant and keeps no global mutable state (forward runs on the Python main thread,
 *    backward on autograd's device thread);
 *  - return value 0 = launched, negative = rejected (message via rat_last_error(), thread-local);
 *  - results are fp32; the default arithmetic of the encoder / head GEMMs is RAT_ARITH_BF16X3 (3-way bf16 split, six of the
 *    nine cross products, fp32 accumulate: fp32-class, not bit-identical to an fp32 FMA chain), RAT_ARITH_F32 selects exact
 *    IEEE fp32 MFMA; everything else (softmax, LayerNorm, GELU, optimizer) is IEEE fp32; ids are int32; token grids are row-major [B][T][S][d]
 *    (sample-in-batch, target||retrieved sample, label||field token, embedding dim).
 */
#ifndef RAT_HIP_H_
#define RAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAT_ABI_VERSION 9
#define RAT_ARITH_F32 0        /* arithmetic selectors of the encoder GEMMs: see rat_attn_fwd_ex */
#define RAT_ARITH_BF16X3 1

int rat_version(void);
const char* rat_last_error(void);

/* One feature field (fuxictr/features.py:46-57 feature_specs entry + its nn.Embedding,
 * fuxictr/pytorch/layers/embedding.py:76-95).  `width` is the table's row length (embedding_dim, or 1 for
 * the LR "wide" tables of fuxictr/pytorch/layers/shallow.py:23-45). */
typedef struct RatField {
    float* table;        /* [vocab][width] weights (forward) or gradient table (backward) */
    int32_t col;         /* first input column (feature_specs[f]["index"]) */
    int32_t ncols;       /* 1 for categorical, max_len for a MaskedSumPooling bag (sequence.py:32-38) */
    int32_t vocab;
    int32_t padding_idx; /* -1 = none; rows with this id receive no gradient (nn.Embedding padding_idx) */
} RatField;

/* ---- K1: token-grid assembly -------------------------------------------------------------------------
 * replaces RAT_m2.forward lines 113-126 (RAT_m2.py): 3x EmbeddingLayer.forward (embedding.py:40-43,
 * 158-178, 138-156), 2x label_embedding_layer, 3x torch.concat.
 * idx [B][T][L] int32, label_ids [B][T] int32 (2 for the target row, the retrieved label otherwise),
 * fields_dev: device array of nfields RatField, label_table [3][d], grid out [B][T][1+nfields][d]. */
int rat_gather_fwd(const int32_t* idx, const int32_t* label_ids, const RatField* fields_dev, int nfields,
                   const float* label_table, float* grid, int B, int T, int L, int d, void* stream);

/* backward of the above (autograd's embedding_dense_backward + cat/stack backward, base_model.py:223).
 * dgrid [B][T][S][d]; dflat (nullable) [B][nfields*d] = gradient of the DNN branch input X_emb.flatten(1)
 * (RAT_m2.py:145-146), added onto the target rows; grad tables are ACCUMULATED into (caller zeroes them);
 * dlabel_table [3][d] is accumulated too. */
int rat_gather_bwd(const float* dgrid, const float* dflat, const int32_t* idx, const int32_t* label_ids,
                   const RatField* grad_fields_dev, int nfields, float* dlabel_table,
                   int B, int T, int L, int d, void* stream);

/* nn.Embedding raises IndexError for an id outside [0, vocab) (embedding.py:158-178 -> F.embedding); the kernels above
 * clamp such ids (and labels outside {0,1,2}) for memory safety only.  rat_check_ids makes the error visible without a
 * per-step host synchronisation: counts[0] += number of feature ids outside their table, counts[1] += number of label ids
 * outside {0,1,2} (device int32[2], caller zeroes it once and reads it at its next synchronisation point). */
int rat_check_ids(const int32_t* idx, const int32_t* label_ids, const RatField* fields_dev, int nfields, int B, int T,
                  int L, int32_t* counts, void* stream);

/* ---- a1 / a2 on the device: inputs_to_device + the label-token rule for a batch that already sits in HBM (ABI v8) --------------
 * base_model.py:125-133, RAT_m2.py:110-118.  X [B][T][L] (x_type: RAT_DTYPE_*, integer-valued) -> idx int32; y [B][T] (fp32 / fp64)
 * -> label_ids int32 (2 in column 0, the retrieved sample's label elsewhere) and y_true fp32 [B] (= y[:, 0]).  One launch. */
#define RAT_DTYPE_I32 0
#define RAT_DTYPE_I64 1
#define RAT_DTYPE_F32 2
#define RAT_DTYPE_F64 3
int rat_batch_prepare(const void* X, int x_type, const void* y, int y_type, int32_t* idx, int32_t* label_ids, float* y_true,
                      int B, int T, int L, void* stream);

/* ---- K0: device-side batch assembly ---------------------------------------------------------------------
 * replaces Dataset.__getitem__ + default_collate + inputs_to_device for retrieval-augmented batches
 * (fuxictr/pytorch/data_generator.py:66-78, 239-241; base_model.py:125-133): the encoded query table
 * data_ids [Q][L] / data_labels [Q], the retrieval pool pool_ids [N][L] / pool_labels [N] and the pre-computed
 * neighbour lists retr_indices [Q][K] (negative = counted from the end of the pool, numpy semantics) are resident
 * in HBM; rows [B] selects the batch.  Outputs are exactly rat_gather_fwd's inputs: idx [B][1+K][L],
 * label_ids [B][1+K] (2 for the target row, the neighbour's label otherwise), plus y_true [B]. */
int rat_batch_assemble(const int32_t* data_ids, const float* data_labels, const int32_t* pool_ids,
                       const float* pool_labels, const int64_t* retr_indices, const int64_t* rows, int32_t* idx,
                       int32_t* label_ids, float* y_true, int64_t Q, int64_t N, int B, int K, int L, void* stream);

/* ---- K2: attention phase of CrossIntraEncoderBlock --------------------------------------------------
 * y = Attention(LayerNorm(x)) + x over groups of L tokens (RAT_m2.py:155-161, 176-202, 222-230).
 * Sequence q in [0, nseq) owns tokens  tok(q, p) = (q / q_div) * hi_stride + (q % q_div) * lo_stride
 * + p * pos_stride, p in [0, L)  (token units; a token is d floats).  Intra-sample attention over the
 * grid: nseq=B*T, L=S, q_div=nseq, lo_stride=S, pos_stride=1.  Cross-sample: nseq=B*S, L=T, q_div=S,
 * hi_stride=T*S, lo_stride=1, pos_stride=S — the reference's reshape/transpose/flatten copy
 * (RAT_m2.py:226-227) becomes this addressing. */
typedef struct RatSeqMap {
    int64_t nseq;
    int32_t L;
    int32_t queries; /* ABI v5 (the padding of v4): 0 or >= L — every position is a query.  0 < queries < L — only the outputs of
                      * positions [0, queries) of each sequence are wanted: the forward may leave the other rows of y / o_save /
                      * lse_save with unspecified (finite) contents, and the backward REQUIRES the dy rows of the other positions
                      * to be zero (their queries contribute nothing; all positions still serve as keys and values, and dx is
                      * complete).  The bf16x3 kernels skip the work; every other kernel computes all positions, which the
                      * contract allows.  Used for the last encoder block, of whose output only x[:, 0][:, 0] is read
                      * (RAT_m2.py:138-140). */
    int64_t q_div, hi_stride, lo_stride, pos_stride;
} RatSeqMap;

typedef struct RatAttnParams {      /* HOST struct of device pointers; state_dict names under          */
    float* ln_g;  /* [d]      encoder.encoder.<i>.<which>_attention.norm.weight                          */
    float* ln_b;  /* [d]      ...norm.bias                                                               */
    float* w_qkv; /* [3*h*dh][d]  ...fn.to_qkv.weight (no bias)                                          */
    float* w_out; /* [d][h*dh]    ...fn.to_out.0.weight, NULL when heads==1 && dim_head==d (Identity)    */
    float* b_out; /* [d]          ...fn.to_out.0.bias                                                    */
    void* planes; /* optional, RAT_ARITH_BF16X3 only: the bf16x3 fragment planes of w_qkv / w_out (rat_attn_planes_bytes bytes,
                   * 16-byte aligned) as written by rat_split_weights_batch from the jobs of rat_attn_split_jobs — valid until
                   * the weights change.  NULL: rat_attn_fwd_ex / rat_attn_bwd_ex derive them themselves (2 - 3 small launches
                   * per call).  Ignored in the struct that receives gradients. */
    const uint64_t* drop_seed_dev; /* ABI v6, optional: DEVICE location of this layer's dropout seed (one of the words rat_dropout_seeds
                   * writes).  When set, rat_attn_fwd_ex / rat_attn_bwd_ex read the seed there and ignore their `dropout_seed` argument:
                   * a training step captured into a hipGraph then draws a new mask on every replay.  NULL: the by-value seed. */
} RatAttnParams;

/* ---- bf16x3 weight planes, once per optimizer step instead of once per call ------------------------------------------------
 * The bf16x3 kernels read every weight matrix as pre-split fragment-major planes.  The planes depend on the
 * weights only, so a training step needs them once: collect the jobs of every layer (rat_attn_split_jobs / rat_ffn_split_jobs
 * fill in what to split where inside that layer's `planes` buffer), run them all in ONE launch (rat_split_weights_batch) and
 * hand the buffers to the forward and backward entry points (RatAttnParams.planes, the `planes` argument of rat_ffn_bwd_res).
 * rat_*_planes_bytes return 0 and rat_*_split_jobs return 0 jobs when no bf16x3 kernel serves the shape. */
typedef struct RatSplitJob {
    const float* w; /* source matrix, row-major with leading dimension ld                                   */
    void* out;      /* fragment planes of B[k][n] = transpose ? w[k*ld + n] : w[n*ld + k], n < N, k < K       */
    int32_t N, K, ld, transpose, perm; /* perm: bit 0 = the k order of stacked accumulator tiles (rat_ffn_split_jobs); bits 8-19 `blk`, bits 20-31
                       * `stride` (ABI v9, 0 = off): the job reads `blk` consecutive ROWS of w out of every `stride` — row i -> (i / blk) * stride +
                       * i % blk — a head group's Q | K | V rows in place (rat_attn_groups_split_jobs) */
    int32_t reserved; /* 0, or n_valid | k_valid << 16: the matrix has only n_valid of the N rows / k_valid of the K columns the planes
                       * cover (0 = all); the rest is written as zeros.  Filled in by rat_*_split_jobs for layers narrower than the
                       * kernels' tiles (embedding_dim 40 / 48 / 56 inside the 64-wide bf16x3 tiles). */
} RatSplitJob;
size_t rat_attn_planes_bytes(int d, int heads, int dim_head);
int rat_attn_split_jobs(const RatAttnParams* w_host, int d, int heads, int dim_head, void* planes, RatSplitJob* jobs_out /* [4] */);
size_t rat_ffn_planes_bytes(int d, int hidden);
int rat_ffn_split_jobs(const float* w1, const float* w2, int d, int hidden, void* planes, RatSplitJob* jobs_out /* [3] */);
int rat_split_weights_batch(const RatSplitJob* jobs_host, int njobs, void* stream);

/* o_save [ntok][h*dh] (attention output before to_out) and lse_save [ntok][h] (log2-domain log-sum-exp of the
 * scaled scores), token-indexed like x, are written when non-NULL (training) and consumed by rat_attn_bwd. */
int rat_attn_fwd(const float* x, float* y, float* o_save, float* lse_save, const RatAttnParams* w_host,
                 const RatSeqMap* map_host, int d, int heads, int dim_head, float ln_eps, void* stream);

size_t rat_attn_bwd_workspace(int d, int heads, int dim_head);
/* dx = dL/dx (overwritten), grads_host->* overwritten with the batch-summed parameter gradients. */
int rat_attn_bwd(const float* x, const float* dy, const float* o_save, const float* lse_save, float* dx,
                 const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace,
                 size_t workspace_bytes, const RatSeqMap* map_host, int d, int heads, int dim_head,
                 float ln_eps, void* stream);

/* The same kernels with the three constants of `PreNorm(Attention)(x) + x` exposed — RAT_m3's block (RAT_m3.py:164-243) runs two
 * attentions on the same input and averages them, with heads/2 heads of width 2*dim_head but the softmax scale of dim_head:
 *   y = out_scale * to_out(softmax(Q K^T * softmax_scale) V) + res
 * res: residual source — x (rat_attn_fwd), y itself (accumulate onto the first attention's result), or NULL;
 * softmax_scale <= 0 selects dim_head^-0.5.  Backward: the gradient through the projection is out_scale * dy and
 * dx = add + LayerNorm-backward(...), add = dy (rat_attn_bwd), another tensor laid out like dx (may alias dx), or NULL.
 *
 * arith selects the arithmetic of the projections (the GEMMs; the softmax core is the same fp32 VALU code either way):
 *   RAT_ARITH_F32     v_mfma_f32_16x16x4_f32: exact fp32, bit-identical to a k-ordered fmaf chain (what rat_attn_fwd / _bwd use);
 *   RAT_ARITH_BF16X3  v_mfma_f32_16x16x32_bf16 on operands split EXACTLY into three bf16 chunks, the six cross products of weight
 *                     >= 2^-16, fp32 accumulation: fp32-class accuracy (the dropped terms are <= 2^-23 of each product — one fp32
 *                     rounding; measured against fp64 in tools/probes/bf16x3_probe.hip) at 2.4x the fp32-MFMA rate.  Compiled for
 *                     the north-star geometry (embedding_dim 64, 8 heads x 10); any other shape runs RAT_ARITH_F32 regardless.
 * The forward needs rat_attn_fwd_workspace() bytes of workspace for the pre-split weight fragments under RAT_ARITH_BF16X3
 * (NULL / too small: exact fp32); the backward's rat_attn_bwd_workspace() already covers its own.
 *
 * dropout_p > 0: the nn.Dropout behind the output projection (Attention.to_out = Sequential(Linear, Dropout), RAT_m2.py:186-189):
 *   y = out_scale * Dropout(to_out(...)) + res, mask(token, column) = a counter-based function of (dropout_seed, token * d + column)
 *   — the generator of rat_dropout — so the backward call re-derives it from the same (p, seed) instead of reading a stored mask
 *   (torch's Philox stream cannot be matched bit for bit; parity for p > 0 is statistical).  Ignored without a projection. */
size_t rat_attn_fwd_workspace(int d, int heads, int dim_head);
int rat_attn_fwd_ex(const float* x, const float* res, float* y, float* o_save, float* lse_save, const RatAttnParams* w_host,
                    const RatSeqMap* map_host, int d, int heads, int dim_head, float softmax_scale, float out_scale,
                    float ln_eps, float dropout_p, uint64_t dropout_seed, int arith, float* workspace, size_t workspace_bytes,
                    void* stream);
int rat_attn_bwd_ex(const float* x, const float* dy, const float* add, const float* o_save, const float* lse_save, float* dx,
                    const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace, size_t workspace_bytes,
                    const RatSeqMap* map_host, int d, int heads, int dim_head, float softmax_scale, float out_scale,
                    float ln_eps, float dropout_p, uint64_t dropout_seed, int arith, void* stream);

/* ---- wide heads in one forward launch (ABI v9).  heads = G x 8 (16 ... 64), dim_head 10, embedding_dim 64 — BASELINE configs[4], the
 * shipped Tmall config's 32 heads (configs/RAT_m2/tmall_x1_002/model_config.yaml:23) at d = 64.  heads * dim_head is too wide for the
 * fused kernels' LDS tile, but the head groups are independent given LayerNorm(x) (RAT_m2.py:192-202: the heads only meet in to_out),
 * so a caller may run the layer as G launches of rat_attn_fwd_ex on 8 heads each (res = x, then res = y) — or as ONE launch that loads
 * and normalises each 64-row chunk once and loops over the groups inside it, adding bias, Dropout and the residual once:
 *   y = out_scale * Dropout(to_out(concat_g softmax(Q_g K_g^T * scale) V_g)) + res        (bf16x3 arithmetic only)
 * planes: rat_attn_groups_planes_bytes() bytes, 16-byte aligned, filled by rat_split_weights_batch from the 4 G jobs of
 *   rat_attn_groups_split_jobs (they read group g's rows of the Q / K / V blocks of to_qkv.weight and its columns of to_out.weight in
 *   place: no permuted copy of the weights) — valid until the weights change; w_host = the layer's full-width parameters.
 * o_save [G][ntok][80], lse_save [G][ntok][8] (or both NULL): group-major, so that slice g is exactly what rat_attn_bwd_ex takes for
 *   a launch on group g; ntok = tokens of x (the slice stride).
 * rat_attn_groups_planes_bytes returns 0 (and rat_attn_groups_split_jobs 0 jobs) for dimensions this form does not serve.
 *
 * The same for the shipped Tmall geometry itself (embedding_dim <= 16, dim_head 10, 16 ... 64 heads; exact fp32, any `arith`): the weights are
 * addressed in place, `planes` is ignored (NULL), and up to 32 heads the BACKWARD is one launch too (rat_attn_bwd_groups): with one
 * 16-wide column tile the weight-gradient accumulators of four head groups fit the registers that one group needs at embedding_dim 64.
 * Gradients arrive in the layer's full-width layout (grads_host = [3*heads*dim_head][d], [d][heads*dim_head], [d], [d], [d]); dx = add +
 * LayerNorm-backward(...) as rat_attn_bwd_ex; workspace: rat_attn_bwd_groups_workspace() bytes (0: not served).
 * Round 6 (same signatures): at embedding_dim <= 16 the groups may also be 4 heads x dim_head 20 (heads = G x 4; RAT_m3 runs num_heads / 2
 * heads of width 2 * dim_head, RAT_m3.py:181): a group's inner width is 80 either way; lse_save is then [G][ntok][4].
 * rat_attn_groups_supported: bit 0 = rat_attn_fwd_groups serves these dimensions, bit 1 = rat_attn_bwd_groups does. */
int rat_attn_groups_supported(int d, int heads, int dim_head);
size_t rat_attn_bwd_groups_workspace(int d, int heads, int dim_head);
int rat_attn_bwd_groups(const float* x, const float* dy, const float* add, const float* o_save, const float* lse_save, int64_t ntok,
                        float* dx, const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace, size_t workspace_bytes,
                        const RatSeqMap* map_host, int d, int heads, int dim_head, float softmax_scale, float out_scale, float ln_eps,
                        float dropout_p, uint64_t dropout_seed, void* stream);
size_t rat_attn_groups_planes_bytes(int d, int heads, int dim_head);
int rat_attn_groups_split_jobs(const RatAttnParams* w_host, int d, int heads, int dim_head, void* planes,
                               RatSplitJob* jobs_out /* [4 * heads / 8] */);   /* slice g of `planes` (rat_attn_groups_planes_bytes() / G each) is also a
                               * valid RatAttnParams.planes for rat_attn_bwd_ex on group g */
int rat_attn_fwd_groups(const float* x, const float* res, float* y, float* o_save, float* lse_save, int64_t ntok,
                        const RatAttnParams* w_host, const void* planes, const RatSeqMap* map_host, int d, int heads, int dim_head,
                        float softmax_scale, float out_scale, float ln_eps, float dropout_p, uint64_t dropout_seed, void* stream);

/* ---- K2d: the attention core alone, for sequences longer than the fused kernel's 64-row tile — RAT_m0 attends jointly over all
 * T*S tokens of a sample (RAT_m0.py:123-127; 231 at the north-star shape).  That variant runs LayerNorm as K2c and the two
 * projections as rat_sgemm; this is softmax(Q K^T * scale) V on the projected rows: qkv [ntok][3*heads*dim_head] (the output of
 * nn.Linear(d, 3I): Q | K | V, head-major), o [ntok][heads*dim_head], lse [ntok][heads] (log2-domain, as rat_attn_fwd saves it);
 * sequence q owns tokens [q*L, (q+1)*L).  softmax_scale <= 0 selects dim_head^-0.5.  dim_head <= 32; one head's K, V (backward:
 * also Q, dO) rows of a sequence must fit LDS (L * dim_head * 16 B <= 160 KB). */
int rat_attn_core_fwd(const float* qkv, float* o, float* lse, int64_t nseq, int L, int heads, int dim_head, float softmax_scale,
                      void* stream);
int rat_attn_core_bwd(const float* qkv, const float* o, const float* lse, const float* dout, float* dqkv, int64_t nseq, int L,
                      int heads, int dim_head, float softmax_scale, void* stream);
/* The same with RatSeqMap addressing (sequences strided through the token grid, e.g. the cross-sample phase) — the composed
 * attention path for dimensions the fused kernels do not serve.  rat_attn_fused_supported tells which path applies: 0 for
 * sequences above 64 tokens or heads*dim_head too wide for the fused LDS tile / weight-gradient accumulators (the shipped
 * Tmall config, configs/RAT_m2/tmall_x1_002/model_config.yaml:23: 32 heads x 10). */
int rat_attn_core_fwd_map(const float* qkv, float* o, float* lse, const RatSeqMap* map_host, int heads, int dim_head,
                          float softmax_scale, void* stream);
int rat_attn_core_bwd_map(const float* qkv, const float* o, const float* lse, const float* dout, float* dqkv,
                          const RatSeqMap* map_host, int heads, int dim_head, float softmax_scale, void* stream);
int rat_attn_fused_supported(int d, int heads, int dim_head, int L);

/* ---- K2 (cont.): FeedForward + residual, y = W2 gelu_erf(W1 x + b1) + b2 + x (RAT_m2.py:163-174, 232) */
int rat_ffn_fwd(const float* x, float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                int64_t ntok, int d, int hidden, void* stream);
size_t rat_ffn_bwd_workspace(int d, int hidden);
int rat_ffn_bwd(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                size_t workspace_bytes, int64_t ntok, int d, int hidden, void* stream);

/* The same block MLP with the residual taken from a SEPARATE tensor: y = W2 gelu_erf(W1 x + b1) + b2 + res (res == x gives
 * rat_ffn_fwd; res == NULL: no residual).  RAT_m1's PreNorm(FeedForward) (RAT_m1.py:143-161,201,207: ff(norm(x)) + x) is
 * rat_layernorm_fwd followed by this with x = norm(x), res = x.  Backward: add_dy = 1 adds dy to dx (residual from x
 * itself), add_dy = 0 returns only the gradient through the two Linear layers.
 * arith: RAT_ARITH_F32 (what rat_ffn_fwd / rat_ffn_bwd use) or RAT_ARITH_BF16X3 (see rat_attn_fwd_ex; compiled for d = 64,
 * hidden = 128, every other shape runs exact fp32). */
int rat_ffn_fwd_res(const float* x, const float* res, float* y, const float* w1, const float* b1, const float* w2,
                    const float* b2, int64_t ntok, int d, int hidden, int arith, void* stream);
int rat_ffn_bwd_res(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                    size_t workspace_bytes, const void* planes /* of rat_ffn_split_jobs, or NULL */, int64_t ntok, int d,
                    int hidden, int add_dy, int arith, void* stream);

/* ABI v7: rat_ffn_bwd_res for an incoming gradient that is zero except on the token rows t = k * dy_period (k = 0, 1, ...): dy_rows
 * [ceil(ntok / dy_period)][d] holds those rows compactly, the zero rows are neither stored nor read.  The last encoder block's case:
 * the head reads one class token per sample (RAT_m2.py:138-140), so d loss / d x is zero on every other token of the [B][T][S] grid and
 * the [ntok][d] zero fill in front of the backward disappears.  Only where rat_ffn_bwd_rows_supported says 1 (the bf16x3
 * weight-stationary kernel: d = 64 or 40 / 48 / 56 with hidden = 2 d, RAT_ARITH_BF16X3); ntok, dy_period < 2^31. */
int rat_ffn_bwd_rows_supported(int d, int hidden, int arith);
int rat_ffn_bwd_res_rows(const float* x, const float* dy_rows, int64_t dy_period, float* dx, const float* w1, const float* b1,
                         const float* w2, const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                         size_t workspace_bytes, const void* planes, int64_t ntok, int d, int hidden, int add_dy, int arith,
                         void* stream);

/* ABI v7 — one reduction launch for a whole backward pass.  rat_attn_bwd* / rat_ffn_bwd* / rat_layernorm_bwd each end with a launch
 * that sums their per-work-group gradient slabs (in `workspace`) into the weight gradients.  Between rat_reduce_defer_begin() and
 * rat_reduce_defer_end(stream, 1) of the SAME host thread those launches are recorded instead and run together at the end (one launch
 * per 80 outputs; same sums in the same order).  Contract: every call in between gets a `workspace` of its own that stays untouched
 * until the end call, and nothing reads the weight gradients before it.  run = 0 drops the record (error unwinding). */
int rat_reduce_defer_begin(void);
int rat_reduce_defer_end(void* stream, int run);

/* ---- K2c: stand-alone nn.LayerNorm(d) (biased variance, affine) over selected token rows — RAT_m1's PreNorm in front of
 * FeedForward and the final `self.norm` of each Transformer (RAT_m1.py:137-141,198,209), of which only token 0 of every
 * sequence is read (RAT_m1.py:125,128).  Row r is read at x + r * x_stride; y is a compact [nrows][d]. */
int rat_layernorm_fwd(const float* x, int64_t x_stride, float* y, const float* gamma, const float* beta, int64_t nrows,
                      int d, float eps, void* stream);
size_t rat_layernorm_bwd_workspace(int64_t nrows, int d);
/* dx row r (at dx + r * dx_stride; rows not addressed are left untouched) = [add row r +] LayerNorm backward of the compact
 * dy [nrows][d]; dgamma / dbeta are OVERWRITTEN with the row sums (fixed-order reduction, no atomics).  `add` (optional)
 * is laid out like dx and may alias it. */
int rat_layernorm_bwd(const float* x, int64_t x_stride, const float* dy, const float* gamma, const float* add, float* dx,
                      int64_t dx_stride, float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                      int64_t nrows, int d, float eps, void* stream);

/* ---- K6: BM25-style top-K retrieval pre-compute — the scoring / top-k / merge core of BM25_topk_retrieval_v4
 * (fuxictr/datasets/data_utils.py:774-1064); without exact-match columns (the shipped dataset configs):
 *   score[b][n] = sum_f (qry_ids[b][f] == db[n][f]) * qry_idf[b][f]   (float64, f ascending)
 *   out_values[b][0:K] = the K largest scores, descending; zero scores are dropped: out_indices = -1, out_values = 0;
 *   out_lens[b] = number of entries kept (data_utils.py:786-818 padded_topk + sort_results).
 * db_ids_field_major is [n_fields][n_db] int32 (the pool's id columns, transposed); qry_ids [n_qry][n_fields] int32; qry_idf
 * [n_qry][n_fields] float64 = log(n_db / count of that id in the pool column), 0 for ids absent from the pool
 * (data_utils.py:873-880,843-847).  Equal scores keep the lower pool index (torch.topk leaves that order unspecified).
 * n_fields <= 32, topk <= 32. */
int rat_bm25_topk(const int32_t* db_ids_field_major, const int32_t* qry_ids, const double* qry_idf, double* out_values,
                  int64_t* out_indices, int64_t* out_lens, int64_t n_db, int64_t n_qry, int n_fields, int topk, void* stream);

/* The same with exact-match columns (`exact_match_col_indices`, data_utils.py:851-866, 932-938): db_groups [n_db] / qry_groups
 * [n_qry] int32 number the distinct values of the exact-match columns (every query group >= 0 — queries whose key the pool
 * does not hold are filtered by the caller, as at data_utils.py:923-924); the id columns passed are the REMAINING ones.
 *   score[b][n] = db_groups[n] == qry_groups[b] ? (BM25 score as above) + 1 : 0 */
int rat_bm25_topk_grouped(const int32_t* db_ids_field_major, const int32_t* db_groups, const int32_t* qry_ids,
                          const double* qry_idf, const int32_t* qry_groups, double* out_values, int64_t* out_indices,
                          int64_t* out_lens, int64_t n_db, int64_t n_qry, int n_fields, int topk, void* stream);

/* ---- K3: prediction head -----------------------------------------------------------------------------
 * Plain fp32 GEMM on MFMA for MLP_Layer's nn.Linear (deep.py:126-141) forward / dgrad / wgrad:
 * C[M][N] = op(A) op(B) (+ bias[N]) (+ beta*C), row-major with leading dimensions, op = transpose flag. */
int rat_sgemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
              float* C, int ldc, const float* bias, float beta, void* stream);
/* Same product with a caller-owned workspace of rat_sgemm_workspace(M, N, K) bytes (0 = none needed): when the output
 * has few tiles and K is long (the weight gradients: K = batch), the k range is split across work-groups and the raw
 * partial tiles are summed in slice order by a second launch — deterministic, no atomics. */
size_t rat_sgemm_workspace(int M, int N, int K);
int rat_sgemm_ws(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, float beta, float* workspace, size_t workspace_bytes, void* stream);
/* ABI v4: the same with the arithmetic selectable — RAT_ARITH_BF16X3 runs the product on v_mfma_f32_16x16x32_bf16 with 3-way split
 * operands (fp32-class accuracy, see rat_attn_fwd_ex) when both operands allow 16-byte fetches and K >= 64; otherwise exact fp32. */
int rat_sgemm_arith(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, float beta, float* workspace, size_t workspace_bytes,
                    int arith, void* stream);
/* BatchNorm1d (train: batch stats, biased var; running stats momentum update with unbiased var; eval:
 * running stats) followed by the hidden layer's activation — deep.py:128-132.  use_bn=0 -> activation only.  z,a [M][N].
 * workspace: rat_bn_workspace(N) bytes (per-row-split partial sums; needed when use_bn && training, and by bwd).
 * act (ABI v6; every rat_bn_* entry point below takes it): the activation MLP_Layer puts behind the layer (deep.py:121-123,
 * torch_utils.get_activation, torch_utils.py:83-94) — the names stay "bn_relu" because ReLU is what every shipped config uses. */
#define RAT_ACT_RELU 0
#define RAT_ACT_NONE 1        /* no activation module (hidden_activations entry None / "") or nn.Identity */
#define RAT_ACT_SIGMOID 2
#define RAT_ACT_TANH 3
#define RAT_ACT_LEAKY_RELU 4  /* nn.LeakyReLU(): negative_slope 0.01 */
#define RAT_ACT_ELU 5         /* nn.ELU(): alpha 1.0 */
size_t rat_bn_workspace(int N);
int rat_bn_relu_fwd(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* save_mean, float* save_rstd, float* workspace, int M, int N,
                    int training, int use_bn, float eps, float momentum, int act, void* stream);
/* a = the forward output (the derivative of every supported activation is a function of it); dgamma/dbeta overwritten */
int rat_bn_relu_bwd(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                    const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, float* workspace,
                    int M, int N, int use_bn, int act, void* stream);
/* ABI v7 — column-strip forms of the two calls above: ONE launch per layer and direction, no workspace, sums in a fixed order
 * (deterministic).  A work-group owns 8 columns for all M rows: batch statistics and their use share the launch, and the backward
 * also returns dbias_lin[N] = column sums of dz — the bias gradient of the nn.Linear in front (deep.py:126-127), i.e. rat_colsum of
 * its own output (nullptr = not wanted).  Need rat_bn_strip_ok(M, N) (N % 4 == 0) and 16-byte aligned pointers; same arithmetic
 * per element as rat_bn_relu_fwd / rat_bn_relu_bwd, batch sums in another (fixed) order. */
int rat_bn_strip_ok(int M, int N);
int rat_bn_act_fwd_strip(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* save_mean, float* save_rstd, int M, int N, int training, int use_bn,
                         float eps, float momentum, int act, void* stream);
int rat_bn_act_bwd_strip(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                         const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, float* dbias_lin,
                         int M, int N, int use_bn, int act, void* stream);
/* The LAST hidden layer in front of the DNN's one-output Linear (deep.py:135-137): its incoming gradient is the outer product
 * da[r][c] = dl[r] * w[c] (dl = dlogit [M], w = that Linear's weight [N]) — formed inside the kernel — and dw[c] = sum_r dl[r] a[r][c]
 * (the Linear's weight gradient) comes back with it: replaces two rat_sgemm launches and the da matrix. */
int rat_bn_act_bwd_strip_outer(const float* z, const float* a, const float* dl, const float* w, float* dw, float* dz,
                               const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                               float* dbias_lin, int M, int N, int use_bn, int act, void* stream);
/* SyncBN for data parallelism (deep.py:128-132 evaluated over the GLOBAL batch, i.e. exactly what the
 * reference's single-device BatchNorm1d sees).  The collectives between the calls belong to the caller (RCCL):
 *   fwd:  rat_bn_local_stats -> all_gather(stats, 2N+1 floats per rank) -> rat_bn_relu_fwd_sync
 *   bwd:  rat_bn_bwd_local_sums -> all_reduce_sum(copy of sums, 2N floats) -> rat_bn_relu_bwd_sync
 * stats = [mean_r (N) | sum (z-mean_r)^2 (N) | row count (1)]; all_stats = [world] such records, combined in rank order
 * (Chan's update: stable, the same bits on every rank).  dgamma / dbeta receive this rank's LOCAL sums (the gradient
 * bucket's all-reduce adds the ranks); the backward reads the global row count from the forward's all_stats.
 * workspace: rat_bn_workspace(N) bytes. */
int rat_bn_local_stats(const float* z, float* stats, float* workspace, int M, int N, void* stream);
int rat_bn_relu_fwd_sync(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* save_mean, float* save_rstd, const float* all_stats, int world,
                         int M, int N, float eps, float momentum, int act, void* stream);
int rat_bn_bwd_local_sums(const float* z, const float* a, const float* da, const float* save_mean, const float* save_rstd,
                          float* sums, float* workspace, int M, int N, int act, void* stream);
int rat_bn_relu_bwd_sync(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                         const float* save_mean, const float* save_rstd, const float* local_sums, const float* global_sums,
                         float* dgamma, float* dbeta, const float* all_stats, int world, int M, int N, int act, void* stream);
/* column sums of a [M][N] matrix (bias gradients); workspace: rat_colsum_workspace(M, N) bytes (= rat_bn_workspace(N) up to
 * M = 65536 rows; more row splits beyond, for the token-sized matrices of the composed attention path) */
size_t rat_colsum_workspace(int M, int N);
int rat_colsum(const float* a, int lda, float* out, float* workspace, int M, int N, void* stream);

/* logit = fc(cls) + dnn_out + sum_f lr_table_f[idx] ; y_pred = sigmoid(logit)  (RAT_m2.py:138-150,
 * shallow.py:36-45); loss_sum += sum_b BCE(y_pred, y_true)/B with torch's log clamp at -100
 * (torch_utils.py:51-63, base_model.py:74-77).  cls rows are read at cls + b*cls_stride (floats); idx rows of
 * the TARGET sample at idx + b*idx_stride.  dnn_out / lr_fields_dev / loss_sum / y_true may be NULL.
 * head (ABI v6): RAT_HEAD_BINARY — the above; RAT_HEAD_REGRESSION — task = "regression" (base_model.py:286-292: no output
 * activation) with loss = "mse_loss": y_pred = logit, loss_sum += sum_b (y_pred - y_true)^2 / B. */
#define RAT_HEAD_BINARY 0
#define RAT_HEAD_REGRESSION 1
int rat_logit_fwd(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b,
                  const float* dnn_out, const RatField* lr_fields_dev, int nfields, const int32_t* idx,
                  int64_t idx_stride, const float* y_true, float* y_pred, float* loss_sum, int B, int d,
                  int head, void* stream);
/* ABI v7: the same with the DNN's one-output Linear evaluated inside: logit = fc(cls) + (dnn_in[b] . dnn_w + dnn_b) + LR — dnn_in
 * [B][dnn_k] with leading dimension dnn_ld is the last hidden layer's output (formerly an N = 1 rat_sgemm launch into dnn_out). */
int rat_logit_fwd_dnn(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b, const float* dnn_in,
                      int64_t dnn_ld, const float* dnn_w, const float* dnn_b, int dnn_k, const RatField* lr_fields_dev,
                      int nfields, const int32_t* idx, int64_t idx_stride, const float* y_true, float* y_pred,
                      float* loss_sum, int B, int d, int head, void* stream);
/* dlogit[b] = gscale * (gscale_dev ? *gscale_dev : 1) * (y_pred - y_true)/B (x 2 for RAT_HEAD_REGRESSION) ; dcls row b (written at
 * dcls + b*dcls_stride) = dlogit*fc_w ; dfc_w, dfc_b and the LR grad tables are ACCUMULATED into (caller zeroes them).
 * gscale_dev (nullable): a DEVICE scalar — autograd's incoming loss gradient — so that backward needs no host read-back. */
int rat_logit_bwd(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride,
                  const float* fc_w, float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w,
                  float* dfc_b, const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx,
                  int64_t idx_stride, float gscale, const float* gscale_dev, int B, int d, int head, void* stream);
/* ABI v7: rat_logit_bwd that also ACCUMULATES sum_b dlogit[b] into ddnn_b (the bias gradient of the DNN's one-output Linear) */
int rat_logit_bwd_dnn(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride, const float* fc_w,
                      float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w, float* dfc_b, float* ddnn_b,
                      const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride, float gscale,
                      const float* gscale_dev, int B, int d, int head, void* stream);

/* ---- K1s: row-sparse / deterministic embedding gradients (BASELINE.json configs[3]; SURVEY.md §2a rows H/I, §7 hard parts 3, 7)
 * replaces, for tables too large for dense semantics, embedding_dense_backward + the dense clip/Adam pass over the tables
 * (embedding.py:158-178 with sparse=False; base_model.py:221-225; torch_utils.py:41-49) and, for every table size, the fp32
 * atomics of rat_gather_bwd by a bit-reproducible sorted segmented reduction.
 *   plan   : (sample, id column) pairs -> key = global row = (fields[f].table - flat_base)/width + id (padding ids, ids outside the
 *            vocabulary and columns with col2field < 0 take no part); stable radix sort; *count_out = number of unique rows U
 *            (device int32; never read back by the library).  target_only: only the t == 0 rows of idx (the LR tables).
 *   reduce : segment sums in sorted (= batch) order -> out_rows [U] int32, out_grads [U][d] (either may be NULL) and / or straight
 *            into a dense gradient block (dense_base + row*d; plain stores, untouched rows are the caller's zeros).
 *            _grid: source rows dgrid [B][T][1+nfields][d] (+ dflat [B][nfields*d] on target rows, as rat_gather_bwd);
 *            _rows: source = gathered lists of `world` ranks, each `cap` entries long with counts_dev[r] valid (all-gather);
 *            _scalar: width-1 tables, source = one value per sample (dlogit [B]; plan built with target_only).
 *   rat_sumsq_rows / rat_adam_rows: gradient-norm contribution and clip + Adam of the U listed rows only ("lazy" Adam: moments
 *            of untouched rows do not decay — declared deviation, exact when every row is touched every step).
 * workspace: rat_sparse_workspace(n) bytes for n = pairs (B*T*L, or B*L with target_only, or cap*world); the same workspace is
 * handed to the matching reduce call. */
/* deterministic gradient of the 3-row label table (RAT_m2.py:64-65): dlabel_table [3][d] += per-block partial sums combined in
 * block order (rat_gather_bwd's version reduces through LDS / global atomics).  workspace: rat_label_grad_workspace(d) bytes. */
size_t rat_label_grad_workspace(int d);
int rat_label_grad(const float* dgrid, const int32_t* label_ids, float* dlabel_table, float* workspace, int64_t nbt, int S,
                   int d, void* stream);
size_t rat_sparse_workspace(int64_t n);
int rat_sparse_plan_ids(const int32_t* idx, const RatField* fields_dev, const int32_t* col2field_dev, int nfields,
                        const float* flat_base, int width, int64_t total_rows, int B, int T, int L, int target_only,
                        void* workspace, size_t workspace_bytes, int32_t* count_out, void* stream);
int rat_sparse_plan_rows(const int32_t* rows, const int32_t* counts_dev, int64_t cap, int world, int64_t total_rows,
                         void* workspace, size_t workspace_bytes, int32_t* count_out, void* stream);
int rat_sparse_reduce_grid(const void* workspace, const int32_t* count_dev, const float* dgrid, const float* dflat,
                           const int32_t* col2field_dev, int B, int T, int L, int nfields, int d, int target_only,
                           int32_t* out_rows, float* out_grads, float* dense_base, void* stream);
int rat_sparse_reduce_rows(const void* workspace, const int32_t* count_dev, const float* src_rows, int64_t cap, int world,
                           int d, int32_t* out_rows, float* out_grads, void* stream);
int rat_sparse_reduce_scalar(const void* workspace, const int32_t* count_dev, const float* per_sample, int B, int L,
                             int32_t* out_rows, float* out_vals, float* dense_base, void* stream);
int rat_sumsq_rows(const float* grads, const int32_t* count_dev, int64_t max_rows, int d, float* norm_sq_out, void* stream);
int rat_adam_rows(float* w_base, float* m_base, float* v_base, const int32_t* rows, const float* grads,
                  const int32_t* count_dev, int64_t max_rows, int d, const float* norm_sq, float max_norm, float lr,
                  float beta1, float beta2, float eps, int step, void* stream);

/* ---- K4/K5: regulariser + clip_grad_norm_ + Adam over flat buffers -----------------------------------
 * base_model.py:79-94,224-225; torch_utils.py:41-49,65-81.
 * rat_l2_reg: for i<n: g[i] += lambda*w[i]; reg_out += (lambda/2)*sum w^2   (the "embedding_layer" tensors);
 *             lambda is multiplied by *lambda_scale_dev when that (device scalar, nullable) is given.
 * rat_sumsq : norm_sq_out += sum g^2 (fp32 partials, fp64-free two-stage tree; caller zeroes the scalar).
 * rat_clip_adam: coef = min(1, max_norm/(sqrt(*norm_sq)+1e-6)); g*=coef; Adam(lr,b1,b2,eps,step), in place. */
int rat_l2_reg(const float* w, float* g, int64_t n, float lambda, const float* lambda_scale_dev, float* reg_out,
               void* stream);
int rat_sumsq(const float* g, int64_t n, float* norm_sq_out, void* stream);
int rat_clip_adam(float* w, const float* g, float* m, float* v, int64_t n, const float* norm_sq,
                  float max_norm, float lr, float beta1, float beta2, float eps, int step, void* stream);

/* ---- ABI v4: the optimizer half of one training iteration in two sweeps, capturable in a hipGraph --------------------
 * base_model.py:79-94 (add_regularization), :221 (clip_grad_norm_), :224 (Adam.step), :220 (zero_grad) — the four dense passes
 * of rat_l2_reg + rat_sumsq + rat_clip_adam + the next step's zero-fill become:
 * rat_sumsq_reg      : *norm_sq_out += sum_i (g[i] + lam(i) w[i])^2, lam(i) = lam_a for i < n_split ("embedding_layer" tensors,
 *                      base_model.py:86) else lam_b (net_regularizer), both multiplied by *lam_scale_dev when given;
 *                      *reg_out (nullable) += sum_i (lam(i)/2) w[i]^2 — the regulariser's VALUE with the unscaled lambdas.
 *                      g is NOT modified: the regulariser gradient only ever exists in registers.
 * rat_clip_adam_fused: coef as rat_clip_adam; g' = (g + lam w) coef; Adam with step size hyper_dev[0] = lr/(1-beta1^t) and
 *                      hyper_dev[1] = 1/sqrt(1-beta2^t); zero_g: g[i] = 0 afterwards (optimizer.zero_grad() of the next step).
 * rat_adam_tick      : the optimizer's clock on the device: *step_dev += 1, hyper_out[0..2] = {lr/(1-beta1^t), 1/sqrt(1-beta2^t),
 *                      lr} from *lr_dev (double arithmetic like torch.optim.Adam's host code) — a replayed graph needs no new
 *                      kernel arguments.
 * rat_adam_rows_dev  : rat_adam_rows with the two step scalars read from hyper_dev. */
int rat_adam_tick(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_out, void* stream);
/* ABI v7 — what a fused training iteration starts with, in one launch: rat_adam_tick, scalars[0 .. nscalars) = 0 (the step's
 * accumulators: BCE sum, clip norm^2, regulariser value) and counters[i] += 1 for i < ncounters (nn.BatchNorm1d.num_batches_tracked of
 * every BatchNorm layer, int64).  Either list may be empty. */
int rat_step_begin(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_out, float* scalars, int nscalars,
                   int64_t* counters, int ncounters, void* stream);
int rat_sumsq_reg(const float* g, const float* w, int64_t n, int64_t n_split, float lam_a, float lam_b,
                  const float* lam_scale_dev, float* norm_sq_out, float* reg_out, void* stream);
int rat_clip_adam_fused(float* w, float* g, float* m, float* v, int64_t n, int64_t n_split, float lam_a, float lam_b,
                        const float* lam_scale_dev, const float* norm_sq, float max_norm, const float* hyper_dev,
                        float beta1, float beta2, float eps, int zero_g, void* stream);
/* ABI v6 — the optimizers torch_utils.get_optimizer (torch_utils.py:41-49) builds besides Adam: getattr(torch.optim, name)(params,
 * lr=lr), i.e. torch's default hyper-parameters.  kind: RAT_OPT_SGD  w -= lr g;  RAT_OPT_ADAGRAD  s += g g, w -= lr g / (sqrt(s) + eps)
 * (eps 1e-10);  RAT_OPT_RMSPROP  s = p0 s + (1 - p0) g g, w -= lr g / (sqrt(s) + eps) (p0 = alpha 0.99, eps 1e-8).  `state` is the one
 * fp32 buffer of the parameters' shape these keep (NULL for SGD).  rat_clip_opt is the counterpart of rat_clip_adam (gradient already
 * holds the regulariser term), rat_clip_opt_fused of rat_clip_adam_fused (g + lambda w formed in registers, g left zero; lr read from
 * hyper_dev[2], which rat_adam_tick keeps current). */
#define RAT_OPT_ADAM 0
#define RAT_OPT_SGD 1
#define RAT_OPT_ADAGRAD 2
#define RAT_OPT_RMSPROP 3
int rat_clip_opt(float* w, const float* g, float* state, int64_t n, const float* norm_sq, float max_norm, float lr, int kind,
                 float p0, float eps, void* stream);
int rat_clip_opt_fused(float* w, float* g, float* state, int64_t n, int64_t n_split, float lam_a, float lam_b,
                       const float* lam_scale_dev, const float* norm_sq, float max_norm, const float* hyper_dev, int kind,
                       float p0, float eps, int zero_g, void* stream);
/* rat_scatter_rows: dense_base[rows[s]*d + c] = grads[s*d + c] for s < *count_dev — the merged (unique rows, gradient rows)
 * lists of all ranks written into the (zeroed) dense gradient block, so that the dense-semantics optimizer (regulariser on every
 * row, base_model.py:79-94) runs unchanged after a row-list exchange. */
int rat_scatter_rows(float* dense_base, const int32_t* rows, const float* grads, const int32_t* count_dev, int64_t max_rows,
                     int d, void* stream);
/* ABI v6 — the same for `lists` lists of capacity `cap` each (rows [lists][cap], grads [lists][cap][d], counts_dev [lists]): what the
 * owner-partitioned exchange of the table gradients leaves on every rank (one reduced list per owner rank, disjoint row ranges). */
int rat_scatter_rows_lists(float* dense_base, const int32_t* rows, const float* grads, const int32_t* counts_dev, int64_t cap,
                           int lists, int d, void* stream);
int rat_adam_rows_dev(float* w_base, float* m_base, float* v_base, const int32_t* rows, const float* grads,
                      const int32_t* count_dev, int64_t max_rows, int d, const float* norm_sq, float max_norm,
                      const float* hyper_dev, float beta1, float beta2, float eps, void* stream);

/* ABI v8 — the owner-partitioned exchange of the table-gradient row lists under data parallelism (the reference
 * has no counterpart — its DataParallel-less loop, base_model.py:213-230, runs on one device; this is the exchange step the sharded
 * path adds).  Rank k owns the rows [k per, (k + 1) per) of a table family.
 *   rat_owner_counts : counts_out[k] = unique rows of a rat_sparse_plan_* plan (workspace, count_dev, n as given to the plan) inside
 *                      owner k's range.  A function of the batch's IDS alone: it runs at the start of the step, the N x N matrix of
 *                      all ranks' counts is all-gathered and copied to the host while the forward runs.
 *   rat_owner_pack   : this rank's sorted lists (A: rows_a / grads_a [.][d]; B, nullable: rows_b / vals_b, width 1) -> `wire`, one
 *                      chunk per owner: [rows A pad 4][gradient rows A][rows B pad 4][values B pad 4] (32-bit words).
 *                      mat_dev: int32 [world][2][world], mat[s][f][k] = rows of family f rank s holds for owner k.
 *   rat_owner_unpack : the chunks received from ranks 0 .. world-1 -> contiguous (rows, gradient rows) pairs in rank order,
 *                      totals[0..1] = pairs received per family; n_extra floats are copied from extra_src to extra_dst.
 *   rat_owner_scatter: `world` all-gathered lists, `stride` words apart, each [count A, count B, -, -][n_extra floats pad 4]
 *                      [rows A: cap_a][gradient rows A: cap_a x d][rows B: cap_b][values B: cap_b] -> dense_a[row * d + c],
 *                      dense_b[row] (plain stores: the owners' ranges are disjoint); extra_out[i] = sum over the lists, in order.
 * max_pairs only sizes the grid. */
int rat_owner_counts(const void* workspace, const int32_t* count_dev, int64_t n, int64_t rows_per_owner, int world,
                     int32_t* counts_out, void* stream);
int rat_owner_pack(const int32_t* mat_dev, int world, int rank, int d, const int32_t* rows_a, const float* grads_a,
                   const int32_t* rows_b, const float* vals_b, int64_t max_pairs, float* wire, void* stream);
int rat_owner_unpack(const int32_t* mat_dev, int world, int rank, int d, const float* wire, int64_t max_pairs, int32_t* rows_a,
                     float* grads_a, int32_t* rows_b, float* vals_b, int32_t* totals, const float* extra_src, float* extra_dst,
                     int n_extra, void* stream);
int rat_owner_scatter(float* dense_a, float* dense_b, float* extra_out, const float* lists, int64_t stride, int world,
                      int64_t cap_a, int64_t cap_b, int d, int n_extra, void* stream);

/* Inverted dropout (nn.Dropout; RAT_m2.py:83,135 emb_dropout, deep.py:133-134 net_dropout): y = keep(seed,i) ? x/(1-p) : 0
 * with a counter-based mask, so the backward pass calls the same function on the gradient with the same seed.
 * In place (y == x) is allowed. */
int rat_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* ABI v6 — the same with the seed read from device memory, and the once-per-training-step refresh of a model's seed words:
 * counter_dev[0] += 1; seeds_dev[i] = mix(base_seed, counter, i), i < n.  nn.Dropout draws from torch's generator on every call
 * (RAT_m2.py:135,186-189, deep.py:133-134); here the generator's state is (base_seed, counter) on the device, so forward and backward
 * of a step agree on the masks (backward reads the same words) and a captured step needs no host-drawn seed. */
/* ABI v6 — FeedForward WITH its two nn.Dropout layers (RAT_m1.py:151-161, RAT_m0.py:150-160: Linear, GELU, Dropout, Linear, Dropout),
 * training mode: y = Dropout2(W2 Dropout1(gelu(W1 x + b1)) + b2) + res, and its backward (dx = [dy +] chain; the residual takes the
 * unmasked dy).  p: the rate both layers share; seed1_dev / seed2_dev: their seed words (rat_dropout_seeds).  Arguments otherwise as
 * rat_ffn_fwd_res / rat_ffn_bwd_res; exact fp32 on the generic kernels (RAT_m2 / RAT_m3 build FeedForward without dropout). */
int rat_ffn_fwd_drop(const float* x, const float* res, float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                     int64_t ntok, int d, int hidden, float p, const uint64_t* seed1_dev, const uint64_t* seed2_dev, void* stream);
int rat_ffn_bwd_drop(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2, const float* b2,
                     float* dw1, float* db1, float* dw2, float* db2, float* workspace, size_t workspace_bytes, int64_t ntok, int d,
                     int hidden, int add_dy, float p, const uint64_t* seed1_dev, const uint64_t* seed2_dev, void* stream);
int rat_dropout_dev(const float* x, float* y, int64_t n, float p, const uint64_t* seed_dev, void* stream);
int rat_dropout_seeds(uint64_t* seeds_dev, int n, uint64_t base_seed, uint64_t* counter_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAT_HIP_H_ */
