/*
 * mhr.h - C ABI of the MI355X (gfx950) hot path for multi-head recommendation:
 * HSTU sequence encoder pieces, item-embedding gather / sparse gradient reduction,
 * sampled-softmax (NCE) loss, full-catalog multi-head scoring + top-k + merge, Adam.
 *
 * The reference (zhykoties/Multi-Head-Recommendation-with-Human-Priors) is pure Python
 * and has no FFI; the boundary it exposes is the model-registry + trainer contract
 * (SURVEY.md section 8b).  This ABI sits one level below that contract: each entry point
 * replaces a group of torch ops inside the reference's HSTU / Collector classes
 * (citations are file:line under code/REC/).  The Python host side
 * (multi-head-recommendation-with-human-priors_amd/ops.py) binds it with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *   - no allocation, no synchronisation inside; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*); calls are re-entrant and graph-capturable;
 *   - return 0 on success, a negative MHR_E* code otherwise (argument checks happen on the
 *     host before launch); mhr_last_error() returns a thread-local message;
 *   - tensors are dense row-major; dtype tags: MHR_F32 = 0, MHR_BF16 = 1;
 *   - item ids are int64 like the reference's batches.
 */
#ifndef MHR_H
#define MHR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHR_F32 0
#define MHR_BF16 1

#define MHR_OK 0
#define MHR_EINVAL (-1)   /* bad argument (null pointer, unsupported size/dtype) */
#define MHR_ELAUNCH (-2)  /* HIP launch failure */

const char* mhr_last_error(void);
int mhr_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Item embedding.
 * ---------------------------------------------------------------------------------------- */

/* Ids outside [0, n_rows) that the gather kernels met since the last reset (they clamp and count where nn.Embedding
 * raises: a kernel cannot).  SYNCHRONISES (device-to-host copy of one word): call it where the host waits anyway.
 * host_count: HOST memory.  reset != 0 clears the counter. */
int mhr_bad_id_count(int64_t* host_count, int reset);

/* out[r,:] = table[ids[r],:]  (nn.Embedding forward, model/IDNet/hstu.py:413,637,670,752,883).
 * Optional fused position add (hstu.py:640-643): when x_out != NULL, ids is viewed as
 * [n_ids/window_len, window_len] and x_out[b,l,:] = table[ids[b,l],:] + pos_table[l,:] for
 * l < seq_len (x_out is [n_x_ids/window_len, seq_len, dim], dtype x_dtype), for the first n_x_ids ids only
 * (n_x_ids <= 0: all of them) - so the item windows and the negative-pool ids of a step are ONE launch.
 * `out` may be NULL when only x_out is wanted.  dim % 4 == 0.  ids outside [0,n_rows) are clamped. */
int mhr_embedding_gather_fwd(const float* table, int64_t n_rows, int dim,
                             const int64_t* ids, int64_t n_ids,
                             void* out, int out_dtype,
                             const float* pos_table, int seq_len, int window_len,
                             void* x_out, int x_dtype, int64_t n_x_ids, void* stream);

/* Everything a TRAINING step reads from the item table in one launch (hstu.py:637-643, 670-672, 752-754):
 *   ids[0, n_item_ids) viewed as [n_item_ids / window_len, window_len] item windows:
 *       rows_out[r,:] = table[ids[r],:] (f32: the targets) and x_out[b,l,:] = table[ids[b,l],:] + pos_table[l,:] for l < seq_len;
 *   ids[n_item_ids, n_ids) - the negative pools:
 *       neg_out[j,:] = table[id,:] / |table[id,:]| as bf16 and neg_norms[j] = |table[id,:]| (f32) - gather + L2 normalisation
 *       without an fp32 copy of the rows; bitwise mhr_embedding_gather_fwd + mhr_l2norm_rows.
 * dim % 4 == 0, dim <= 2048; ids outside [0, n_rows) are clamped and counted (mhr_bad_id_count). */
int mhr_embedding_gather_step(const float* table, int64_t n_rows, int dim, const int64_t* ids, int64_t n_ids,
                              int64_t n_item_ids, float* rows_out, const float* pos_table, int seq_len, int window_len,
                              float* x_out, void* neg_out, float* neg_norms, void* stream);

/* Dense embedding backward (ATen embedding_dense_backward): grad_table[ids[r],:] += grad_rows[r,:]
 * with float atomics.  grad_table [n_rows, dim] f32 must be zeroed by the caller. */
int mhr_embedding_scatter_add_bwd(const void* grad_rows, int grad_dtype, const int64_t* ids, int64_t n_ids,
                                  float* grad_table, int64_t n_rows, int dim, void* stream);

/* Sparse embedding backward, deterministic.  Inputs: ids sorted ascending (`sorted_ids`) with `perm`
 * giving, for every sorted position, the source row in the caller's gradient rows
 * (sorted_ids = ids[perm]).  Source rows come from up to two buffers laid end to end:
 * rows [0, n_a) from grad_a, rows [n_a, n_a+n_b) from grad_b.  Optional `x_grad`
 * ([n_a/window_len, seq_len, dim] f32): the gradient of the fused position-added copy, added to
 * source row r < n_a when (r % window_len) < seq_len.
 * Output: for every segment head position i (first occurrence of an id) out_rows[i,:] = sum of the
 * segment's source rows (f32), and row_slot[id] = i.  out_rows ([n_ids, dim] f32) must be ZEROED by the caller;
 * non-head rows are zero on return.  No atomics: a segment cut by the 32-position work chunks is summed from one
 * partial per chunk in chunk order by a second launch, so the result is bitwise reproducible (and identical on every
 * data-parallel replica).  row_slot ([n_rows] int32) must hold -1 everywhere on entry; mhr_adam_rows restores that.
 * Ids outside [0, n_rows) get no slot (their rows are left out of the update). */
int mhr_sparse_rows_segment_sum(const int64_t* sorted_ids, const int64_t* perm, int64_t n_ids,
                                const void* grad_a, int a_dtype, int64_t n_a,
                                const void* grad_b, int b_dtype, int64_t n_b,
                                const float* x_grad, int seq_len, int window_len,
                                float* out_rows, int32_t* row_slot, int64_t n_rows, int dim, void* stream);

/* Dense AdamW over an embedding table whose gradient is given sparsely (trainer.py:292-299 semantics:
 * every row is updated every step, untouched rows with g = 0).  For each row: slot = row_slot[row];
 * g = slot >= 0 ? grad_rows[slot,:] * grad_scale : 0; AdamW update; row_slot[row] = -1.
 * step is 1-based.  If `row_slot` is NULL, grad_rows is a dense [n_rows, dim] gradient. */
int mhr_adam_rows(float* w, float* m, float* v, int64_t n_rows, int dim,
                  const float* grad_rows, int32_t* row_slot, float grad_scale,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* Dense AdamW over a flat parameter buffer (all non-table parameters live in one flat buffer).  w_bf16 (optional, [n]):
 * bf16 copy of the updated weights, written in the same pass - the operand of the next step's bf16 GEMMs, so no
 * per-step cast kernels (the reference's autocast re-casts every weight every step).
 * step_dev (optional, device int64[1]) + hist [hist_len, 4] f32 (rows of mhr_adam_consts): a step replayed from a hipGraph
 * has its kernel arguments frozen at capture, so the step's constants (lr decay, bias corrections) are read from
 * hist[step_dev[0] % hist_len] instead of being derived from `lr` / `step`; NULL: the arguments as given. */
int mhr_adam_flat(float* w, const float* g, float* m, float* v, int64_t n, float grad_scale,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* w_bf16,
                  const float* hist, int hist_len, const int64_t* step_dev, void* stream);
/* Lazy form of mhr_adam_rows (same arithmetic, bitwise the same weights once a row is up to date): rows are replayed through
 * the gradient-free steps they missed when they are next needed.  last_step [n_rows] int32 (zeros at start): last step applied
 * to a row; hist [hist_len, 4] f32: the constants of step s at row s % hist_len as filled by mhr_adam_consts (a host helper,
 * no launch); the caller flushes (mode 2) at least every hist_len steps.  mode 0: bring the rows of `ids` (duplicates allowed) up
 * to step - 1 (before the forward reads them); mode 1: ids = the sorted ids of a SparseRowGrad, grad_rows its rows: every
 * segment head is replayed to step - 1 and gets step `step` with its gradient (row_slot of the row is reset to -1); mode 2: all
 * rows through `step`.  step_dev (optional, device int64[1]): overrides `step` (hipGraph-replayed steps). */
int mhr_adam_consts(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* out4);
int mhr_adam_rows_lazy(float* w, float* m, float* v, int64_t n_rows, int dim, const int64_t* ids, int64_t n_ids,
                       const float* grad_rows, int32_t* row_slot, int32_t* last_step, const float* hist,
                       int hist_len, int step, float grad_scale, float beta1, float beta2, float eps, int mode,
                       const int64_t* step_dev, void* stream);

/* out[c] += sum_r x[r, c]: x bf16 [rows, cols] (cols % 8 == 0), out fp32 [cols].  The reduction of split-K weight-gradient
 * partials and of bias gradients straight into the flat gradient buffer (autograd's accumulate semantics: the caller
 * zeroes the buffer once per step).  Many-row inputs are reduced by several workgroups meeting through float atomics. */
int mhr_sum_rows_into(const void* x_bf16, int64_t rows, int64_t cols, float* out, void* stream);
/* The same with f32 rows (the position-table gradient: column sums of d_x [B, L D] over the batch, hstu.py:640-643's backward). */
int mhr_sum_rows_f32_into(const float* x, int64_t rows, int64_t cols, float* out, void* stream);
/* The same column sums for n equally shaped bf16 matrices in ONE launch: ptrs is a DEVICE int64 array of 2 n addresses - the n
 * sources [rows, cols] first, then the n fp32 destinations [cols] (accumulated into, float atomics across row ranges). */
int mhr_sum_rows_many(const int64_t* ptrs, int n, int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation / gating (HSTU layer, model/IDNet/hstu.py:213-219, 241, 277-285).
 * ---------------------------------------------------------------------------------------- */

/* Affine-free LayerNorm over the last dim: y = (x - mean) * rstd; saves mean/rstd ([rows] f32, may be NULL). */
int mhr_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, float* mean, float* rstd,
                      int64_t rows, int dim, float eps, void* stream);
/* dx (+)= rstd * (dy - mean(dy) - xhat * mean(dy*xhat));  accumulate != 0 adds into dx (f32 only). */
int mhr_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                      void* dx, int dx_dtype, int accumulate, int64_t rows, int dim, void* stream);
/* out = x + y, out_bf16 = bf16(out); x / out f32 [n], y / out_bf16 bf16 [n], n % 8 == 0, 16-byte aligned.  The residual add
 * behind the LAST encoder layer (model/IDNet/hstu.py:286-288), whose sum the decoding heads read in f32 (their residual) and in
 * bf16 (their GEMM operand, llm_heads.py:30-40 under autocast); its backward is the same call on (d_out, d_out_bf16). */
int mhr_add_cast(const float* x, const void* y_bf16, float* out, void* out_bf16, int64_t n, void* stream);

/* Residual add fused with the next layer's LayerNorm (model/IDNet/hstu.py:286-287 then 241):
 *   x_out = x + y,  xn = LN(x_out)      x, x_out f32 [rows, dim]; y, xn bf16; mean / rstd [rows] saved for the backward.
 * Backward: total = d_xout + LN'(d_xn)  written as dx (f32, gradient of x) and dy (bf16, gradient of y). */
/* first_row / seq_len (optional, here and in mhr_ln_gate_*): rows = B * seq_len token rows of front-padded sequences and
 * first_row[b] = index of sequence b's first valid key (mhr_attn_seq_layout).  Rows in front of it - except a sequence's last
 * row, which the decode reads - take part in nothing (no valid key reads them, loss and decode skip them, every gradient that
 * reaches them is exactly zero): their operands are not loaded but read as zeros and the zeros the arithmetic produces are
 * written.  Live rows keep every bit; NULL: all rows live.  (The half-wave form of mhr_ln_gate_* - dim 256, bf16 - ignores it:
 * latency-bound one-shot waves gain nothing from skipped loads.) */
int mhr_add_layernorm_fwd(const float* x, const void* y_bf16, float* x_out, void* xn_bf16, float* mean, float* rstd,
                          int64_t rows, int dim, float eps, const int32_t* first_row, int seq_len, void* stream);
int mhr_add_layernorm_bwd(const void* d_xn_bf16, const float* x_out, const float* mean, const float* rstd,
                          const float* d_xout, float* dx, void* dy_bf16, int64_t rows, int dim,
                          const int32_t* first_row, int seq_len, void* stream);

/* o = silu(u) * LayerNorm(a) * dropmask   (hstu.py:277-285).  u is a column block of the uvqk GEMM
 * output: u[r,c] = u_base[r*u_stride + c] (pre-activation; SiLU applied here).  a [rows, dim].
 * dropout: keep-mask from a counter hash of (seed, r*dim+c), scaled by 1/(1-p); p = 0 disables.
 * step_seed (optional, device int64[1]): the step counter of a hipGraph-replayed step, whose kernel arguments are frozen
 * at capture: the effective seed is (step_seed[0] * 1000003 + seed) & (2^63 - 1), NULL: `seed` as given. */
int mhr_ln_gate_fwd(const void* u_base, int64_t u_stride, const void* a, int dtype, void* o, int o_dtype,
                    float* mean, float* rstd, int64_t rows, int dim, float eps,
                    float dropout_p, uint64_t seed, const int64_t* step_seed,
                    const int32_t* first_row, int seq_len, void* stream);
/* Backward of the above: given d_o, writes du (pre-activation gradient, into a column block with
 * row stride du_stride) and da. */
int mhr_ln_gate_bwd(const void* d_o, int do_dtype, const void* u_base, int64_t u_stride, const void* a, int dtype,
                    const float* mean, const float* rstd, void* du_base, int64_t du_stride, void* da,
                    int64_t rows, int dim, float dropout_p, uint64_t seed, const int64_t* step_seed,
                    const int32_t* first_row, int seq_len, void* stream);

/* y[r,:] = x[r,:] / ||x[r,:]||_2  (hstu.py:605-606, 672, 966, 975, 1021); optional norms out ([rows] f32). */
int mhr_l2norm_rows(const void* x, int x_dtype, void* y, int y_dtype, float* norms, int64_t rows, int dim, void* stream);
/* backward of y = x / |x| (autograd of hstu.py:672, 754): dx = (dy - n (n . dy)) / |x| with n = x / |x|; all f32, norms from the forward */
int mhr_l2norm_rows_bwd(const float* dy, const float* x, const float* norms, float* dx, int64_t rows, int dim, void* stream);
/* Backward of y = table[ids] / |table[ids]| w.r.t. the gathered rows, reading them THROUGH the id list (the forward is the
 * negative-pool half of mhr_embedding_gather_step: no fp32 copy of the gathered rows exists; the table is unchanged between a
 * step's forward and backward).  dx = (dy - n (n . dy)) / |x|, all f32; ids outside [0, n_src_rows) are clamped. */
int mhr_l2norm_rows_indexed_bwd(const float* dy, const float* table, int64_t n_src_rows, const int64_t* ids, const float* norms,
                                float* dx, int64_t rows, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * HSTU pointwise-gated attention (model/IDNet/hstu.py:137-160), fused, never materialises [L,L].
 *   out[b,n,h,:] = sum_{m<=n, key_valid[b,m]} silu(silu(q)[b,n,h].silu(k)[b,m,h]) / L * silu(v)[b,m,h,:]
 * q,k,v are column blocks of the pre-activation uvqk GEMM output ([B*L, row_stride] bf16): the inner
 * SiLU of hstu.py:244-245 is applied on load (apply_silu != 0).  head_dim = dqk = dv, multiple of 8, <= 128.
 * key_valid [B,L] uint8.  out [B*L, n_heads*head_dim] bf16.  Any L <= 131072: short sequences keep the K / V tile
 * images of one (batch, head) resident in LDS (one workgroup per (batch, head)); when the images would fill more than
 * 64 KiB the streamed form runs instead (one workgroup per block of 128 queries, K / V tiles through a two-slot LDS
 * ring: csrc/attention_stream.hip).  Same arithmetic, same bits.  MHR_ATTN_STREAM=0/1 in the environment forces the form
 * where both can run.
 * ---------------------------------------------------------------------------------------- */
int mhr_hstu_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride,
                      const uint8_t* key_valid, void* out,
                      void* act_q, void* act_k, void* act_v, int64_t act_stride,
                      int B, int L, int n_heads, int head_dim, int apply_silu, void* stream);
/* act_q/act_k/act_v (optional, all or none): the forward also stores the activated operands
 * (silu(q), silu(k), silu(v), bf16, row stride act_stride, same head layout) for the backward.
 *
 * Backward: given d_out ([B*L, n_heads*head_dim] bf16) writes the gradients w.r.t. the PRE-activation
 * q,k,v (chain rule through the load-time SiLU included when apply_silu != 0) into column blocks with row
 * stride d_stride (bf16).  act_* are the activated operands saved by the forward (pass q,k,v themselves
 * when apply_silu == 0) - or all NULL with apply_silu != 0: silu(q), silu(k), silu(v) are then recomputed from the
 * pre-activation inputs while they are staged (the forward need not store them: 3 B*L*D bf16 less to write and to read).
 * Two passes (dK/dV by key block, dQ by query block), no atomics: bitwise reproducible.  Resident form (all four operand
 * images in LDS, one workgroup per (batch, head)) while 4 images fit 80 KiB, streamed form (two launches, Q/dO resp.
 * K/V tiles through the LDS ring) beyond that. */
int mhr_hstu_attn_bwd(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride,
                      const void* act_q, const void* act_k, const void* act_v, int64_t act_stride,
                      const uint8_t* key_valid, const void* d_out,
                      void* dq, void* dk, void* dv, int64_t d_stride,
                      int B, int L, int n_heads, int head_dim, int apply_silu, void* stream);

/* Token-rows projection of the encoder layers: C[M,N] (bf16) = A[M,K] (bf16, row stride lda) . W^T + bias[N] (bf16, optional),
 * fp32 accumulation - reference model/IDNet/hstu.py:236-239 (`torch.mm(normed_x, self._uvqk)`), hstu.py:281-288
 * (`self._o(...)`) and their input gradients, under bf16-mixed autocast.  W is stored [N,K] (w_is_kn = 0: an nn.Linear
 * weight in its forward) or [K,N] (w_is_kn = 1: the `_uvqk` parameter; an nn.Linear weight in its input gradient), row
 * stride ldw.  K in {64, 128, 256} (the stationary operand lives in registers), N a multiple of 8 (of 256 with w_is_kn),
 * leading dimensions multiples of 8 elements, 16-byte aligned operands; mhr_rows_gemm_supported answers whether a shape is
 * covered (the caller uses the library GEMM otherwise, as for every other dense projection). */
int mhr_rows_gemm_supported(int M, int N, int K, int w_is_kn);
int mhr_rows_gemm(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, const void* bias,
                  void* c, int64_t ldc, int M, int N, int K, void* stream);

/* Sequence layout of a batch of masks, computed once per batch and shared by every layer's attention launches:
 * first_block[b] = the 32-row block holding sequence b's first valid key (ceil(L/32) when it has none); seq_order (optional) =
 * the sequences sorted by it, most live blocks first (stable).  The reference's loaders pad at the FRONT (data/dataset/
 * trainset.py:111-137, evalset.py:34-41), so blocks before first_block take part in nothing: their outputs and gradients are
 * exactly zero.  The *_seq entries take both arrays (either may be NULL): the resident kernels then skip staging, tile pairs
 * and epilogues of the dead blocks (writing the zeros) and run the sequences in seq_order; results are bit-identical to the
 * plain entries, which are the *_seq entries with both NULL. */
int mhr_attn_seq_layout(const uint8_t* key_valid, int B, int L, int32_t* first_block, int32_t* seq_order,
                        int32_t* first_row /* optional: index of the first valid key, L when none */, void* stream);
int mhr_hstu_attn_fwd_seq(const void* q, const void* k, const void* v, int64_t row_stride,
                          const uint8_t* key_valid, void* out,
                          void* act_q, void* act_k, void* act_v, int64_t act_stride,
                          int B, int L, int n_heads, int head_dim, int apply_silu,
                          const int32_t* first_block, const int32_t* seq_order, const int32_t* cu_rows, int n_rows_total,
                          void* stream);
int mhr_hstu_attn_bwd_seq(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride,
                          const void* act_q, const void* act_k, const void* act_v, int64_t act_stride,
                          const uint8_t* key_valid, const void* d_out,
                          void* dq, void* dk, void* dv, int64_t d_stride,
                          int B, int L, int n_heads, int head_dim, int apply_silu,
                          const int32_t* first_block, const int32_t* seq_order, const int32_t* cu_rows, int n_rows_total,
                          void* stream);
/* cu_rows (optional, int32 [B + 1]): PACKED sequences - the rows of sequence b are cu_rows[b] .. cu_rows[b + 1] - 1 of the
 * operands (the valid positions of the B windows back to back, no padding rows; key_valid then covers the packed rows); L stays
 * the window length (LDS budget, the 1 / n of hstu.py:158) and n_rows_total the row count of the operands: the rows behind the last
 * sequence are written as zeros.  Resident form only. */

/* Packed token rows (csrc/rows_pack.hip): the valid positions of the B windows back to back in a [capacity, dim] buffer, so
 * that the encoder's layers (hstu.py:221-328) run over the valid rows only - the loaders' front padding (trainset.py:111-137,
 * evalset.py:34-41) is 37 % of the rows of the synthetic cfg1 batch.  mhr_seq_pack_maps numbers the valid positions sequence by
 * sequence in window order: cu_rows [B + 1] (sequence b owns packed rows cu_rows[b] .. cu_rows[b + 1] - 1), src_of [capacity]
 * (window row b L + l of a packed row, -1 behind the last sequence), row_of [B L] (packed row of a window position, -1 for
 * padding).  capacity is the caller's static bound (a bucketed count known to the loader); positions past it are dropped and
 * overflow[0] (optional) receives the real count, 0 otherwise.  mhr_rows_gather_masked: out[r] = idx[r] >= 0 ? src[idx[r]] : 0
 * (f32 or bf16 rows) - pack, unpack and both their backward passes (the map is injective). */
int mhr_seq_pack_maps(const uint8_t* key_valid, int B, int L, int capacity, int32_t* cu_rows, int32_t* src_of, int32_t* row_of,
                      int32_t* overflow, void* stream);
int mhr_rows_gather_masked(const void* src, int dtype, const int32_t* idx, void* out, int64_t n_out, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * LLM decoder blocks of the HLLM twin: the user decoder `user_llm(inputs_embeds=...)`
 * (model/HLLM/hllm.py:501-502, 781-783) and the item tower (hllm.py:399-464) are Llama-style stacks
 * (model/HLLM/modeling_llama.py:729-795, baichuan/modeling_baichuan.py:110-338).  The dense projections stay library
 * GEMMs; everything between them is here.
 * ---------------------------------------------------------------------------------------- */

/* RMSNorm (modeling_llama.py:266-280), optionally fused with the residual add in front of it (779/783, 785/768):
 *   x_out = x (+ res);  y = bf16(weight * x_out * rsqrt(mean(x_out^2) + eps));  rstd [rows] saved for the backward.
 * x, x_out f32 [rows, dim]; res, y bf16; weight f32 [dim].  res == NULL: plain norm of x (x_out ignored).  dim <= 4096. */
int mhr_rmsnorm_fwd(const float* x, const void* res_bf16, const float* weight, float* x_out, void* y_bf16, float* rstd,
                    int64_t rows, int dim, float eps, void* stream);
/* dx = rstd * (dy*w - x_hat * mean(dy*w*x_hat)) (+ d_xout, the residual path's gradient, may be NULL) as f32 and, when
 * dres_bf16 != NULL, the same values as bf16 (the gradient of the fused residual branch).  The weight gradient
 * sum_rows dy * x_hat is returned as dw_part [mhr_rmsnorm_bwd_parts(rows), dim] f32 partials (deterministic; the caller
 * sums dim 0). */
int mhr_rmsnorm_bwd_parts(int64_t rows);
int mhr_rmsnorm_bwd(const void* dy_bf16, const float* x_out, const float* weight, const float* rstd, const float* d_xout,
                    float* dx, void* dres_bf16, float* dw_part, int64_t rows, int dim, void* stream);

/* SwiGLU gate of the MLP (modeling_llama.py:484; xformers swiglu in modeling_baichuan.py:197-206) on the output of one
 * [gate | up] GEMM:  act[r, c] = silu(gate_up[r, c]) * gate_up[r, ffn + c].  All bf16; ffn % 8 == 0. */
int mhr_swiglu_fwd(const void* gate_up_bf16, void* act_bf16, int64_t rows, int ffn, void* stream);
int mhr_swiglu_bwd(const void* gate_up_bf16, const void* d_act_bf16, void* d_gate_up_bf16, int64_t rows, int ffn, void* stream);

/* Rotary position embedding, in place, rotate_half convention (modeling_llama.py:426-441), on `n_heads` consecutive
 * heads of width head_dim at the start of each row of x (bf16, row stride in elements): q and k of a packed qkv row
 * in one call.  positions [n_tokens] int32 (NULL: token t sits at t % seq_len); cos/sin tables [max_pos, head_dim/2]
 * f32.  inverse != 0 applies the transposed rotation (the backward). */
int mhr_rope_inplace(void* x_bf16, int64_t row_stride, const int32_t* positions, const float* cos_table,
                     const float* sin_table, int64_t n_tokens, int seq_len, int n_heads, int head_dim, int max_pos,
                     int inverse, void* stream);

/* Causal softmax attention with grouped KV heads (modeling_llama.py:648-682; flash path 683-705 ->
 * flash_self_attn.py:61-130).  q/k/v are column blocks of one token matrix with a common row stride (elements): query
 * head h at q + h*head_dim, KV head g at k / v + g*head_dim, g = h / (n_heads / n_kv_heads).  Sequence b is the row range
 * [cu_seqlens[b], cu_seqlens[b+1]) (packed `cu_input_lens` batches) or, with cu_seqlens == NULL, [b*max_len, (b+1)*max_len);
 * every sequence must be <= max_len rows.  key_valid [n_tokens] uint8 (NULL: all valid) masks padding keys.
 *   out [n_tokens, n_heads*head_dim] bf16,  lse [n_tokens, n_heads] f32 (natural log, scaled scores; 0 for rows with no
 *   admissible key, whose output is 0).  K and V of one sequence stay in LDS: max_len * head_dim <= ~40k.
 * Backward: dq into a matrix with row stride dq_stride (query head h at column h*head_dim); dk / dv hold ONE SLAB PER
 * QUERY HEAD ([n_tokens, n_heads*head_dim], row stride dkv_stride) - the caller sums each group of n_heads/n_kv_heads
 * slabs into its KV head.  No atomics, bitwise reproducible. */
int mhr_softmax_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const int32_t* cu_seqlens,
                         const uint8_t* key_valid, void* out, float* lse, int n_seqs, int max_len, int n_heads,
                         int n_kv_heads, int head_dim, float scale, void* stream);
int mhr_softmax_attn_bwd(const void* q, const void* k, const void* v, int64_t row_stride, const int32_t* cu_seqlens,
                         const uint8_t* key_valid, const void* out, const void* d_out, const float* lse, void* dq,
                         int64_t dq_stride, void* dk, void* dv, int64_t dkv_stride, int n_seqs, int max_len, int n_heads,
                         int n_kv_heads, int head_dim, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Token compaction (replaces the boolean-mask indexing of model/IDNet/hstu.py:688-690, 814-829 and its host-side
 * `mask.sum() == 0` branch).  mask [n_groups, n_slots] uint8 (0/1) marks the live (group, slot) pairs; q_all
 * [n_groups, n_slots], p_all / o_all [n_slots] int32 are static tables (query row, target row, prediction offset
 * of a slot).  For every group the live slots are written in ascending slot order:
 *   q_idx / p_idx / o_idx [n_groups, tok_cap][j] = table value of the j-th live slot,  n_tok[g] = min(#live, tok_cap)
 * (entries >= n_tok[g] are left untouched).  scratch: [n_groups, ceil(n_slots / 4096)] int32.  Two launches.
 * ---------------------------------------------------------------------------------------- */
int mhr_token_compact(const uint8_t* mask, const int32_t* q_all, const int32_t* p_all, const int32_t* o_all,
                      int n_groups, int n_slots, int tok_cap, int32_t* q_idx, int32_t* p_idx, int32_t* o_idx,
                      int32_t* n_tok, int32_t* scratch, int32_t* tok_of_slot, void* stream);
/* tok_of_slot (may be NULL) [n_groups, n_slots] int32: the inverse map - position of a live slot in its group's list,
 * -1 for slots that are not live (or fell beyond tok_cap). */

/* ------------------------------------------------------------------------------------------
 * Decoding heads after their one concatenated GEMM (model/llm_heads.py:5-40, stacked and permuted by hstu.py:665-667):
 *   out[b, h, l, :] = x[b, l, :] + silu(z[b, l, h, :]),  x [n_tok, dim] f32 (n_tok = B seq_len), z [n_tok, n_heads dim] bf16,
 *   out [B, n_heads, seq_len, dim] f32 - the layout the loss reads.  Backward: dz = d_out silu'(z) (bf16), dx = sum_h d_out (f32).
 * ---------------------------------------------------------------------------------------- */
int mhr_heads_residual_fwd(const float* x, const void* z_bf16, float* out, int64_t n_tok, int seq_len, int n_heads,
                           int dim, void* stream);
int mhr_heads_residual_bwd(const float* d_out, const void* z_bf16, void* dz_bf16, float* dx, int64_t n_tok,
                           int seq_len, int n_heads, int dim, void* stream);

/* Row maps of compacted token lists (query-row sharing): a row = a run of consecutive live tokens with the same q_idx.
 * q_idx [n_groups, tok_cap], n_tok_dev [n_groups] -> tok2row [n_groups, tok_cap] (0 beyond n_tok), row_q / row_first
 * [n_groups, row_cap] (query row and first token of row r; row_first[n_row] = n_tok; rows >= n_row untouched: the caller
 * zeroes them), n_row [n_groups].  scratch: [n_groups, ceil(tok_cap / 4096)] int32.  row_cap > the largest row count.  Two launches. */
int mhr_row_maps(const int32_t* q_idx, const int32_t* n_tok_dev, int n_groups, int tok_cap, int row_cap,
                 int32_t* row_q, int32_t* row_first, int32_t* tok2row, int32_t* n_row, int32_t* scratch, void* stream);

/* Training-time log counters of one group (hstu.py:621-629): over the live tokens of prediction offset 0 (o_idx == 0) of
 * `group`, out[0] = mean n_valid ("nce_samples"), out[1 + i] = mean(rank < ks_host[i]) ("nce_top{k}_acc"), i < n_k <= 5.
 * n_valid / rank / o_idx [n_groups, tok_cap] int32; ks_host is HOST memory (read at launch).  out [1 + n_k] f32.
 * scratch8: 8 x uint64 on the device, all zero at the first launch; every launch leaves it zero again (the workgroups'
 * integer partial sums and a completion ticket live there). */
int mhr_nce_log_counters(const int32_t* n_valid, const int32_t* rank, const int32_t* o_idx, const int32_t* n_tok_dev,
                         int group, int tok_cap, const int32_t* ks_host, int n_k, uint64_t* scratch8, float* out, void* stream);

/* Weighted total of the per-(group, offset) mean losses plus the logged partial sums in ONE launch (reference hstu.py:697-723,
 * 836-870: `loss_p = mean over the offset's tokens`, horizon discount x head weight, per-segment and per-head sums).
 * bucket_sum / bucket_cnt [G, P] are mhr_nce_finalize's; weight [G, P]; P % n_segments == 0.
 * total [1]; out [G P + G S + G + S] = per_gp = sum / max(cnt, 1) * weight | seg_all [G, S] | g_tot [G] | seg_sum [S] (over groups).
 * Sums in index order (deterministic).  _bwd: w_out [G P] = d_total[0] * weight / max(cnt, 1), the per-bucket token weight
 * mhr_nce_shared_bwd_rows / mhr_nce_bwd_tokens take. */
int mhr_loss_reduce(const float* bucket_sum, const float* bucket_cnt, const float* weight, int n_groups, int n_buckets,
                    int n_segments, float* total, float* out, void* stream);
int mhr_loss_reduce_bwd(const float* d_total, const float* bucket_cnt, const float* weight, int n, float* w_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sampled softmax with false-negative suppression (model/IDNet/hstu.py:600-619 + F.cross_entropy 697/833).
 * Tokens are described by row indices instead of compacted copies: token t uses query row
 * q_rows[q_idx[t]] and positive row p_rows[p_idx[t]] (both raw, L2-normalised inside; io_dtype f32/bf16).
 * negs [n_neg, dim] bf16 are already L2-normalised.  n_tok is read from device memory (*n_tok_dev)
 * so data-dependent token counts need no host sync; `tok_cap` bounds the grid.  dim in {16,32,64,128,256}.
 * GROUPS: one launch serves n_groups independent (token list, negative pool) problems - the prior categories of a
 * step - laid out along a leading axis: q_idx/p_idx/w/lse/loss/... [n_groups, tok_cap], n_tok_dev [n_groups],
 * negs [n_groups, round_up(n_neg, 32), dim] (each pool padded to whole 32-row tiles with finite rows, e.g. zeros; the
 * padding rows never contribute), qn/pn [n_groups, tok_cap, dim], supp [n_groups, ceil(n_neg/32), tok_cap],
 * d_negs [n_groups, n_neg, dim] (unpadded); q_rows / p_rows / dq_rows / dp_rows are shared row spaces.
 *   logit_pos = s*cos(q,p); logit_j = s*cos(q,neg_j), dropped when cos(p,neg_j) > thres; s = exp(clamp(logit_scale,0,ln 100))
 *   loss[t] = logsumexp(logits) - logit_pos;  lse[t] saved for backward (both written by mhr_nce_finalize).
 * Optional logs (may be NULL): n_valid[t] = 1 + #kept negatives; rank[t] = #kept negatives with logit > logit_pos
 * (hstu.py:621-629: nce_samples and top-k accuracy follow from these).
 * Saved for backward (may be NULL when no backward follows): qn_out/pn_out [tok_cap, dim] bf16 normalised rows,
 * supp_out [ceil(n_neg/32), tok_cap] uint32 (bit j of word [t, tok] = negative 32t+j suppressed for that token),
 * q_inv/p_inv [tok_cap] = 1/||row||, s_pos [tok_cap] = cos(q,p).
 * fix_words (optional, with u_out): scratch [n_groups, ceil(n_neg/32), round_up(n_p_rows, 256)] uint32, n_p_rows = rows of
 * p_rows.  The false-negative test cos(p, neg) > thres depends on the TARGET ROW only, and a row is the positive of many
 * tokens (every (position, offset) pair that points at it): with fix_words the test runs once per (group, row, negative)
 * into this bit table (a second launch, in front) and the fused forward carries the query operand only.  Same results.
 * fix_row_list [n_groups, round_up(n_p_rows, 256)] int32 + fix_n_rows [n_groups] int32 (device) + fix_slot_of_row
 * [n_groups, n_p_rows] int32 scratch (optional, together): the rows of p_rows that tokens of each group can point at
 * (any superset); only those are tested, column j of the bit table = list entry j, and the inverse map is written to
 * fix_slot_of_row.  Rows outside the list must not be referenced by live tokens of that group.
 * Plain form (u_out set; supp_out, fix_words and the fix_* lists NULL; pn_out optional; n_neg % 32 == 0): the fused
 * forward with NOTHING suppressed - no bit table, no suppression words loaded or stored, no bit test per logit.  This is
 * what the query-row-sharing path launches on its row list (the per-token kernels take the false negatives back out).
 * ---------------------------------------------------------------------------------------- */
int mhr_nce_fwd(const void* q_rows, const int32_t* q_idx, const void* p_rows, const int32_t* p_idx, int io_dtype,
                const void* negs, int n_neg, int dim, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                const float* logit_scale_dev, float thres,
                float* sum_out, int32_t* n_valid, int32_t* rank,
                void* qn_out, void* pn_out, uint32_t* supp_out, float* q_inv, float* p_inv, float* s_pos,
                int log_group, float* u_out, int64_t n_p_rows, uint32_t* fix_words, const int32_t* fix_row_list,
                const int32_t* fix_n_rows, int32_t* fix_slot_of_row, void* stream);
/* log_group: the one group whose n_valid / rank are wanted (-1 = every group); the other groups skip the counting.
 * u_out (may be NULL; needs every saved tensor and tok_cap % 32 == 0): the TRAINING path.  [n_groups, tok_cap, dim] f32,
 *   u_out[g, t, :] = sum_j keep_tj exp(scale (s_tj - 1)) negs[g, j, :]
 * the unnormalised token-side gradient: d(loss_t)/d(qn_t) restricted to the negatives is exp(scale - lse_t) * scale *
 * u_out[t].  It is accumulated by the forward itself (the gated tile goes straight back into the matrix pipe against
 * the LDS-resident negatives), so the backward needs no second pass over the negatives for the token side. */
/* The forward is split over negative ranges (grid.y) so that (token block, negative range) units fill the chip
 * without a tail: each unit adds its partial sum_j keep*exp(scale*(s_j - 1)) into sum_out[t] (and its counts into
 * n_valid / rank) with atomics - the caller zeroes sum_out, n_valid, rank - and mhr_nce_finalize produces
 * lse[t] = scale + log(sum[t] + exp(scale*(s_pos[t]-1))), loss[t] = lse[t] - scale*s_pos[t], n_valid[t] += 1. */
int mhr_nce_finalize(const float* sum, const float* s_pos, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                     const float* logit_scale_dev, float* loss, float* lse, int32_t* n_valid,
                     const int32_t* bucket_idx, int n_buckets, float* bucket_sum, float* bucket_cnt, void* stream);
/* Optional bucket sums (bucket_idx may be NULL): bucket_idx [n_groups, tok_cap] int32 in [0, n_buckets) names the
 * prediction offset of each token; bucket_sum / bucket_cnt [n_groups, n_buckets] f32 (caller zeroes) receive
 * sum(loss) and the token count per (group, offset) - the reference takes the MEAN loss per offset before weighting
 * (hstu.py:697-700, 833-836), so no per-token tensor has to go back through an index_add. */
/* Backward, two kernels (one launch each).  w[t] = d(total loss)/d(loss[t]) (0 for unused slots); inputs are the
 * forward's saved tensors.
 * mhr_nce_bwd_tokens: row-wise.  From u (mhr_nce_fwd's u_out) it forms dQn_t = w_t exp(scale - lse_t) u_t, adds the
 *   positive-pair term, applies the L2-normalisation chain rule and accumulates the gradient w.r.t. the RAW query /
 *   positive rows into dq_rows[q_idx[t], :] / dp_rows[p_idx[t], :] (f32, same row spaces as the forward's q_rows /
 *   p_rows; float atomics because several tokens share a row; caller zeroes); atomically adds d(logit_scale parameter)
 *   into *d_logit_scale (may be NULL); writes lw_out[t] = lse[t] log2(e) - log2(w[t]) ([n_groups, tok_cap] f32, may be
 *   NULL), the per-token exponent offset mhr_nce_bwd_negs consumes: w exp(scale s - lse) = exp2(scale log2(e) s - lw).
 *   w: per-token [n_groups, tok_cap] when w_bucket is NULL; otherwise w is [n_groups, n_buckets] and token t of group g
 *   weighs w[g, w_bucket[g, t]] (the gradient of a per-offset mean is constant inside a bucket).
 * mhr_nce_bwd_negs: negative-stationary streaming GEMM; recomputes the logits, replays the false-negative decisions
 *   from `supp`, and accumulates d_negs ([n_neg, dim] f32, float atomics across token splits; caller zeroes) w.r.t.
 *   the normalised negatives.  `lw` is the array mhr_nce_bwd_tokens wrote (launch that first).  Requires
 *   tok_cap % 32 == 0: the forward pads the last live 32-token tile of the saved state (zero qn / pn rows, all-ones
 *   suppression words) so that whole token tiles stream without clamping.  supp == NULL: nothing was suppressed (the
 *   plain form of mhr_nce_fwd); lw of the padding tokens of the last live tile must then be +inf (mhr_nce_row_lw writes
 *   +inf behind the live rows; callers of mhr_nce_shared_bwd_rows pre-fill lw_row with +inf). */
int mhr_nce_bwd_tokens(const void* qn, const void* pn, const float* u, int dim, int n_groups,
                       const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev,
                       const float* lse, const float* w, const float* q_inv, const float* p_inv, const float* s_pos,
                       const int32_t* q_idx, const int32_t* p_idx,
                       float* dq_rows, float* dp_rows, float* d_logit_scale, float* lw_out,
                       const int32_t* w_bucket, int n_buckets, int64_t* dq_fix, int64_t* dp_fix,
                       float* dls_part, void* stream);
int mhr_nce_bwd_negs(const void* qn, const void* negs, const uint32_t* supp, int n_neg, int dim, int n_groups,
                     const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev,
                     const float* lw, float* d_negs, int64_t* dn_fix, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sampled softmax with QUERY-ROW SHARING (csrc/nce_shared.hip; same reference lines as mhr_nce_fwd, plus the window
 * construction hstu.py:682-690, 808-829 that makes P tokens share one query).  The tokens (b, l, p), p = 0..P-1, of a
 * prior category use the same query row and differ in their target only, so the negative-pool products are evaluated per
 * DISTINCT ROW: the caller runs mhr_nce_fwd and mhr_nce_bwd_negs on the row list (plain form: nothing suppressed)
 * and these entry points add what depends on the token.  Token lists must keep the tokens of a row adjacent.
 *   tok2row [n_groups, tok_cap] int32: row of token t;  row arrays are [n_groups, row_cap(, dim)], token arrays
 *   [n_groups, tok_cap(, dim)];  row_first [n_groups, row_cap] int32: first token of row r (entry n_row = n_tok).
 * mhr_nce_fix_bits: the false-negative bit table of mhr_nce_fwd as a launch of its own (same arguments), plus
 *   fix_any [n_groups, round_up(n_p_rows, 256)] int32 (caller zeroes): != 0 where a target row has any suppressed negative;
 *   bit k = a suppressed negative in the 32-negative tiles [k << s, (k + 1) << s), s the smallest shift with ceil(n_tiles / 2^s) <= 32
 *   (the token kernels read only the flagged groups of the row's column of the bit table).
 * The normalised targets are a property of the TARGET ROW: pn_rows [n_p_rows, dim] bf16 = bf16(p_rows / |p_rows|) and p_inv
 *   [n_p_rows] f32 (mhr_l2norm_rows, once per step) are shared by all tokens and groups; `pn` / `p_inv` below are these tables.
 * mhr_nce_shared_fwd_tokens: per token: s_pos = qn_row . pn_rows[p_idx],
 *   sum_tok = sum_row - sum over the token's suppressed negatives of exp(scale (s_j - 1)); with log counters:
 *   n_valid_tok = n_valid_row - #suppressed, rank_tok = rank_row - #suppressed with s_j > s_pos (rank_row must have been
 *   counted against this token's target: the caller gives the row kernel the target of the row's first token).
 *   Then mhr_nce_finalize on the token arrays as usual.
 * mhr_nce_shared_bwd_tokens: mhr_nce_bwd_tokens reading qn / u / q_inv of the token's ROW; takes the token's suppressed
 *   negatives out of u (bf16-rounded, as the row kernel accumulated them) and subtracts their share from d_negs
 *   ([n_groups, n_neg, dim] f32, may be NULL); writes lw_out per token.
 * mhr_nce_row_lw: lw_row[r] = -log2 sum_{t in r} 2^(-lw_tok[t]) (the weight of a row in mhr_nce_bwd_negs); rows beyond
 *   n_row are left untouched (the caller fills lw_row with +inf).
 * ---------------------------------------------------------------------------------------- */
int mhr_nce_fix_bits(const void* p_rows, int io_dtype, int64_t n_p_rows, const void* negs, int n_neg, int dim,
                     int n_groups, float thres, uint32_t* fix_words, const int32_t* fix_row_list,
                     const int32_t* fix_n_rows, int32_t* fix_slot_of_row, int32_t* fix_any, void* stream);
int mhr_nce_shared_fwd_tokens(const void* pn_rows, int64_t n_p_rows, const int32_t* p_idx,
                              const int32_t* tok2row, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                              int row_cap, const void* qn_row, const float* sum_row, const int32_t* n_valid_row,
                              const int32_t* rank_row, const void* negs, int n_neg, int dim,
                              const float* logit_scale_dev, const uint32_t* fix_words,
                              const int32_t* fix_slot_of_row, const int32_t* fix_any,
                              float* s_pos, float* sum_tok, int32_t* n_valid_tok, int32_t* rank_tok, void* stream);
int mhr_nce_shared_bwd_tokens(const void* qn_row, const float* u_row, const float* q_inv_row, int row_cap,
                              const int32_t* tok2row, const void* pn, int dim, int n_groups,
                              const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev,
                              const float* lse, const float* w, const float* p_inv, const float* s_pos,
                              const int32_t* q_idx, const int32_t* p_idx, float* dq_rows, float* dp_rows,
                              float* d_logit_scale, float* lw_out, const int32_t* w_bucket, int n_buckets,
                              const void* negs, int n_neg, const uint32_t* fix_words, int64_t n_p_rows,
                              const int32_t* fix_slot_of_row, const int32_t* fix_any, float* d_negs, void* stream);
/* Window-structured lists (slots (b, l, p), p fastest; target row of a slot = b (seq_len + pred_len) + l + 1 + p; tok_of_slot
 * from mhr_token_compact): the backward without per-token atomics.
 * mhr_nce_shared_bwd_rows: a half-wave per (group, row): dq_rows[row_q[r]] += the row's gradient (chain rule once per row),
 *   lw_row[r] written, d(logit_scale) added, suppressed pairs taken out of u and of d_negs.  exclusive_rows != 0: the caller
 *   guarantees that no two (group, row) pairs name the same query row (the groups read disjoint decoding heads) - the add is
 *   then a plain read-modify-write; otherwise one float-atomic set per row.
 * mhr_nce_shared_bwd_targets: one wave per row of p_rows: dp_rows[m] += the gradient of every token of every group that
 *   points at it, gathered through tok_of_slot (plain read-modify-write, no atomics, bitwise reproducible). */
int mhr_nce_shared_bwd_rows(const void* qn_row, const float* u_row, const float* q_inv_row, const int32_t* row_q,
                            const int32_t* row_first, const int32_t* n_row_dev, int row_cap, const void* pn, int dim,
                            int n_groups, int tok_cap, const float* logit_scale_dev, const float* lse,
                            const float* w, const float* s_pos, const int32_t* p_idx, float* dq_rows,
                            float* d_logit_scale, float* lw_row, const int32_t* w_bucket, int n_buckets,
                            const void* negs, int n_neg, const uint32_t* fix_words, int64_t n_p_rows,
                            const int32_t* fix_slot_of_row, const int32_t* fix_any, float* d_negs, int exclusive_rows, int64_t* dn_fix, float* dls_part, void* stream);
int mhr_nce_shared_bwd_targets(const void* qn_row, int row_cap, const int32_t* tok2row, const int32_t* tok_of_slot,
                               const int32_t* n_tok_dev, int n_groups, int n_slots, int tok_cap, int seq_len,
                               int pred_len, const void* pn, const float* p_inv, int dim,
                               const float* logit_scale_dev, const float* lse, const float* w, const float* s_pos,
                               const int32_t* w_bucket, int n_buckets, int64_t n_p_rows, float* dp_rows,
                               void* stream);
int mhr_nce_row_lw(const float* lw_tok, const int32_t* row_first, const int32_t* n_row_dev, int n_groups,
                   int tok_cap, int row_cap, float* lw_row, void* stream);

/* ------------------------------------------------------------------------------------------
 * Full-catalog multi-head scoring + top-k + cross-head merge
 * (model/IDNet/hstu.py:965-1015, trainer/trainer.py:724-726, evaluator/collector.py:241-282).
 * ---------------------------------------------------------------------------------------- */

/* Streaming scorer: for every row r of `users` ([n_rows, dim] bf16, L2-normalised head embeddings,
 * row r = b*H + h) and every item n in {item_begin + j*item_stride < n_items}: s = users[r].items[n]
 * (bf16 MFMA, f32 accumulate).  The candidate (s, n) is appended to row r's list when
 *   s >= tau[r]  &&  (tag_bits[n] & row_bits[r]) != 0  &&  n != 0  &&  n not in history(b = r / H).
 * tag_bits [n_items] uint32: bit c set when item n belongs to category c, bit 31 always set;
 * row_bits [n_rows] uint32: the bit the row requires (1<<c, or 1<<31 for unconstrained rows, 0 disables the row).
 * history: CSR over users: hist_ptr [n_rows/H + 1] int32, hist_items sorted ascending per user (may be NULL).
 * Lists: cand_val/cand_idx [n_rows, cap]; cand_cnt [n_rows] int32 (caller zeroes) counts all qualifying
 * candidates, including those beyond cap (overflow is detected by cand_cnt > cap). */
int mhr_catalog_score_emit(const void* users, int n_rows, int H, const void* items, int64_t n_items, int dim,
                           int64_t item_begin, int64_t item_stride,
                           const uint32_t* tag_bits, const uint32_t* row_bits, const float* tau,
                           const int32_t* hist_ptr, const int64_t* hist_items,
                           float* cand_val, int32_t* cand_idx, int32_t* cand_cnt, int cap, void* stream);

/* fp32 re-score of candidate lists: out_val[r, j] = users[r, :] . items[cand_idx[r, j], :] in fp32 for j < cand_cnt[r]
 * (users [n_rows, dim], items [n_items, dim] fp32, cand_idx [n_rows, k2] int64), out_idx = the indices as int32: the list
 * format of mhr_topk_select.  The scorers rank bf16 operands; the reference ranks fp32 scores (hstu.py:965-979,
 * collector.py:245): |s_bf16 - s_fp32| <= 2^-8 for unit vectors, so selecting on the re-scored candidates within 2^-7 of
 * the k-th bf16 score gives the reference's fp32 top-k (ops.catalog_topk_exact). */
int mhr_rescore_f32(const float* users, const float* items, int dim, int64_t n_items, const int64_t* cand_idx, int n_rows,
                    int k2, const int32_t* cand_cnt, float* out_val, int32_t* out_idx, void* stream);

/* Dense scores of a FEW rows, every score kept (model/IDNet/hstu.py:965-1015 fp32 score matmul + tag masks,
 * trainer/trainer.py:724-726 pad / history suppression - what the reference does for every row on its [B,H,N] tensor):
 * for the n_list rows r = row_list[j] of users [*, dim]: out_val[j, n] = users[r, :] . items[n, :] accumulated in fp32
 * (operands of `dtype`: MHR_F32 or MHR_BF16, users and items alike), or -inf where item n is the pad id 0, fails
 * (tag_bits[n] & row_bits[r]) != 0 (tag_bits NULL: bit 31 for every item) or is in the history of user r / H (CSR hist_ptr
 * int32 / hist_items int64 sorted per user; both NULL: no filter); out_idx[j, n] = n, out_cnt[j] = n_items: the list format
 * of mhr_topk_select with cap = n_items, which ranks the row exactly (value desc, index asc).  The exact re-run of rows the
 * threshold scorers cannot certify (ops.catalog_topk_exact, wide.py).  dim % 4 == 0, dim <= 4096; out_* [n_list, n_items]. */
int mhr_catalog_score_rows_dense(const void* users, const void* items, int dtype, int dim, int64_t n_items,
                                 const int32_t* row_list, int n_list, int H, const uint32_t* tag_bits,
                                 const uint32_t* row_bits, const int32_t* hist_ptr, const int64_t* hist_items,
                                 float* out_val, int32_t* out_idx, int32_t* out_cnt, void* stream);

/* The decode's exactness bookkeeping (ops.catalog_topk / catalog_topk_exact; replaces the masks the reference needs none of because
 * it ranks a dense [B, H, N] score tensor, hstu.py:965-1015 + collector.py:245) - one launch per decision:
 *  pick_tau:     tau[r] = kth2[r] if both sample selects were clean (status 0) and kth2 is finite, else kth1[r] if clean, else -inf.
 *  flag:         flagged[r] = status != 0 || (count < k_min && row_bits != 0 && isfinite(tau)); any_out[0] = 1 if any row is (written).
 *  margin_count: count[r] = finite entries of the sorted list sorted_vals[r, 0..k2) that are >= sorted_vals[r, kk - 1] - margin.
 *  uncertified:  full[r] = (count >= k2 && list_can_fill) || (kth finite && tau != -inf && !(kth - margin >= tau)), kth = sorted_vals[r, kk - 1]
 *                (tau = -inf: an exact list, nothing was left below it; a NaN or +inf tau certifies nothing);
 *                any_out[0] = 1 if any row is (written).  flag / uncertified run as ONE workgroup (n_rows = users x heads). */
int mhr_topk_pick_tau(const float* kth1, const float* kth2, const int32_t* st1, const int32_t* st2, int n_rows, float* tau, void* stream);
int mhr_topk_flag(const int32_t* status, const int32_t* count, const int32_t* row_bits, const float* tau, int k_min, int n_rows,
                  uint8_t* flagged, int32_t* any_out, void* stream);
int mhr_topk_margin_count(const float* sorted_vals, int n_rows, int k2, int kk, float margin, int32_t* count, void* stream);
int mhr_topk_uncertified(const int32_t* count, const float* sorted_vals, int k2, int kk, const float* tau, int list_can_fill,
                         float margin, int n_rows, uint8_t* full, int32_t* any_out, void* stream);

/* Fast path of the same scorer ("sliced lists").  Every (row, item slice, lane half) triple owns a short list
 *   cand_val / cand_idx [n_rows, 2 n_slices, cap_s],  cand_cnt [n_rows, 2 n_slices] (written, no zeroing needed)
 * whose fill count lives in a register of the lane that owns it (mhr_topk_select_sliced then takes 2 n_slices lists): a threshold hit costs two plain stores (no atomic,
 * no global load - the tile's tag words travel to LDS with the tile).  The history filter is NOT applied here but in
 * mhr_topk_select_sliced.  n_slices: positive multiple of 8 (item tiles are split evenly over the slices; grid =
 * ceil(n_rows / 256) * n_slices workgroups).  items_alloc_rows: rows of `items` that may be read; when item_stride == 1
 * and the catalog is readable to the end of its last 32-row tile the tiles stream unclamped. */
int mhr_catalog_score_emit_sliced(const void* users, int n_rows, const void* items, int64_t n_items, int64_t items_alloc_rows,
                                  int dim, int64_t item_begin, int64_t item_stride, const uint32_t* tag_bits,
                                  const uint32_t* row_bits, const float* tau, float* cand_val, int32_t* cand_idx,
                                  int32_t* cand_cnt, int n_slices, int cap_s, void* stream);
/* Exact top-k over a row's sliced lists (same order and completion rule as mhr_topk_select), dropping the items of the
 * user's history (CSR hist_ptr / hist_items as above, may be NULL; user = row / H).  count_out[row] = candidates left
 * after the history filter; status[row] = 1 when a slice list (count > cap_s) or the on-chip array (8192 keys)
 * overflowed - such rows must be re-run exactly by the caller. */
int mhr_topk_select_sliced(const float* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int n_slices, int cap_s,
                           int n_rows, int H, const int32_t* hist_ptr, const int64_t* hist_items, int k,
                           float* out_val, int64_t* out_idx, float* kth_val, int32_t* count_out, int32_t* status, void* stream);

/* Per-row exact selection of the k best candidates: value descending, index ascending on ties.
 * Rows with fewer than k candidates are completed with (-inf, lowest item ids not in the list).
 * out_val/out_idx [n_rows, k]; kth_val [n_rows] (optional) receives the k-th value.
 * status[r] (optional) = 1 when cand_cnt[r] > cap (list truncated: result not trustworthy). */
int mhr_topk_select(const float* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int cap,
                    int n_rows, int k, float* out_val, int64_t* out_idx, float* kth_val, int32_t* status,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Wide-feature path (feature dim > 256: HSTU size-4, the HLLM twin): the logit contractions run as bf16 x bf16 -> f32
 * library GEMMs over token / item chunks; these are the fused one-pass epilogues over one f32 chunk.
 * ---------------------------------------------------------------------------------------- */

/* Sampled-softmax epilogue (model/IDNet/hstu.py:600-629, 697) over rows [row_base, row_base + rows) of one group:
 * neg_logits / fix_logits [rows, ld] f32 = cos(query, negative) / cos(target, negative); s_pos [rows] = cos(query, target);
 * scale_dev[0] = exp(clamped logit_scale); negatives with fix > thres are suppressed.  Per row:
 *   lse = scale + log(sum_kept exp(scale (s - 1)) + exp(scale (s_pos - 1))),  loss = lse - scale s_pos,
 *   n_valid = #kept + 1, rank = #kept with s > s_pos   (n_valid / rank may be NULL).
 * Rows at or beyond n_live_dev[0] (NULL: all live) get loss = 0, n_valid = rank = 0. */
int mhr_nce_dense_fwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* s_pos,
                      const float* scale_dev, float thres, const int32_t* n_live_dev, int64_t row_base, int64_t rows,
                      float* lse, float* loss, int32_t* n_valid, int32_t* rank, void* stream);
/* Gradient tile g[r, j] = w[r] exp(scale s[r, j] - lse[r]) keep[r, j] (0 on rows that are not live), bf16 [rows, ldg]:
 * the operand of dQ = g . N and dN = g^T . Q. */
int mhr_nce_dense_bwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* lse,
                      const float* w, const float* scale_dev, float thres, const int32_t* n_live_dev, int64_t row_base,
                      int64_t rows, void* g_bf16, int64_t ldg, void* stream);
/* REMI's interest-aware hard-negative loss over the same dense chunks (model/IDNet/remi.py:203-288): with l = scale s,
 *   loss = logaddexp(l+, A - (Z - log n_neg)) - l+,  A = logsumexp_j((beta + 1) l_j),  Z = logsumexp_j(beta l_j)  over the kept
 * negatives (beta > 0; beta <= 0 is mhr_nce_dense_fwd).  Saves lse (the logaddexp), log_num = A, log_imp = Z per row; the
 * backward writes g = w sigma_neg ((beta + 1) exp((beta + 1) l - A) - beta exp(beta l - Z)) as the bf16 tile of the dQ / dN GEMMs. */
int mhr_ihn_dense_fwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* s_pos,
                      const float* scale_dev, float thres, float beta, const int32_t* n_live_dev, int64_t row_base,
                      int64_t rows, float* lse, float* log_num, float* log_imp, float* loss, int32_t* n_valid,
                      int32_t* rank, void* stream);
int mhr_ihn_dense_bwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* lse,
                      const float* log_num, const float* log_imp, const float* w, const float* scale_dev, float thres,
                      float beta, const int32_t* n_live_dev, int64_t row_base, int64_t rows, void* g_bf16, int64_t ldg,
                      void* stream);

/* The same scorer at feature dims beyond the register-stationary kernels (dim a multiple of 64, up to 8192: HSTU size-4,
 * the HLLM twin's TinyLlama / Baichuan2 widths; hstu.py:965-1015 = hllm.py:838-883): an LDS-tiled MFMA GEMM (256 items x
 * 128 rows per workgroup, 64-feature chunks by LDS-DMA, three stages, loader + consumer waves) with the threshold emit in
 * its epilogue - the score block never exists in memory.  Operands are PACKED tile images (mhr_pack_tiles): the selected
 * item rows {item_begin + j * item_stride} with tiles_per_block = 8, the user rows with tiles_per_block = 4; a (256-item
 * block, 64-feature chunk) is then one contiguous 32 KB run.  Same predicates and list format as
 * mhr_catalog_score_emit_sliced with FOUR lists per (row, item slice):
 *   cand_val / cand_idx [n_rows, 4 n_slices, cap_s], cand_cnt [n_rows, 4 n_slices] (written; no zeroing).
 * A list sees at most 64 items per 256-item block.  n_slices must be mhr_catalog_wide_slices(n_rows) (32 ... 256: the
 * kernel arranges row blocks x item slices per XCD so that both operands' chunks are shared in that XCD's L2).
 * item_begin / item_stride only name the items (candidate ids, tag words); the packed image holds them densely. */
int mhr_catalog_score_emit_wide(const void* users_packed, int n_rows, const void* items_packed, int64_t n_items, int dim,
                                int64_t item_begin, int64_t item_stride, const uint32_t* tag_bits, const uint32_t* row_bits,
                                const float* tau, float* cand_val, int32_t* cand_idx, int32_t* cand_cnt, int n_slices,
                                int cap_s, void* stream);
int mhr_catalog_wide_slices(int n_rows);

/* Sampled-softmax logit contractions at feature dims beyond 256 (dim a multiple of 64, up to 8192; model/IDNet/hstu.py:600-619
 * = model/HLLM/hllm.py:378-397 + F.cross_entropy hstu.py:697, 833; log counters hstu.py:621-629): the wide scorer's LDS-tiled
 * MFMA GEMM (256 negatives x 128 tokens per workgroup, packed tile images, loader + consumer waves) with the loss arithmetic in
 * its epilogue - no [N_tok, n_neg] logit block in memory.  Operands: mhr_pack_tiles images of L2-normalised bf16 rows - the
 * token rows (queries or targets, [n_rows, dim]) with tiles_per_block = 4, the negatives ([n_neg, dim]) with tiles_per_block = 8.
 * Per-token arrays hold t_pad = ceil(n_rows / 128) * 128 entries; n_tiles = ceil(n_neg / 256) * 8.
 *   fix_bits:   bits [n_tiles * 2, t_pad] uint16: bit g of word ((tile * 2 + lane half) * t_pad + token) = cos(target, negative) > thres
 *               for the g-th negative that accumulator layout gives the lane (the two consumers below read the same layout).
 *   fwd:        per token tot = sum_j keep exp(scale (s_j - 1)), #kept, #{kept s_j > s_pos} as partials [n_lists, t_pad]
 *               (n_lists = 4 * mhr_catalog_wide_slices(n_rows); plain stores, no atomics), then in the same call
 *               lse = scale + log(tot + exp(scale (s_pos - 1))), loss = lse - scale s_pos (0 beyond n_live_dev[0]),
 *               n_valid = #kept + 1, rank (both optional).  bits NULL: nothing suppressed.  scale_dev: device scalar exp(logit_scale).
 *   grad_tile:  g [n_rows, ldg] bf16 = keep * w[token] * exp(scale s - lse[token]) (0 beyond n_live_dev[0] and for padding
 *               negatives): the operand of the plain gradient products dQ = G N and dN = G^T Q.  ldg % 4 == 0, ldg >= n_neg. */
int mhr_nce_wide_fix_bits(const void* targets_packed, int n_rows, const void* negs_packed, int n_neg, int dim, float thres,
                          uint16_t* bits, void* stream);
int mhr_nce_wide_fwd(const void* queries_packed, int n_rows, const void* negs_packed, int n_neg, int dim, const uint16_t* bits,
                     const float* s_pos, const float* scale_dev, const int32_t* n_live_dev, float* part_tot, int32_t* part_nv,
                     int32_t* part_rk, float* lse, float* loss, int32_t* n_valid, int32_t* rank, void* stream);
/* The two plain products of that backward on the same core (no library GEMM left in the loss):
 *   out[r, i] (+)= alpha_dev[0] * sum_k A[i, k] B[r, k]   (fp32 accumulation of bf16 operands, out fp32 [n_r, ldc])
 * a_packed: n_i rows packed with tiles_per_block = 8, b_packed: n_r rows with tiles_per_block = 4, both with k_dim (a multiple of
 * 64) contraction entries per row; dQ = G N: A = N^T (mhr_pack_tiles_t of the negatives), B = G; dN = G^T Q: A = Q^T, B = G^T.
 * mhr_pack_tiles_t packs an operand stored TRANSPOSED: packed row j = column j of x [n_k, ld] bf16 (contraction index down the
 * rows; padded with zeros to whole 64-entry chunks): mhr_pack_tiles_bytes(n_sel, round_up(n_k, 64), tiles_per_block) bytes. */
int mhr_wide_gemm_nt(const void* a_packed, int n_i, const void* b_packed, int n_r, int k_dim, const float* alpha_dev, float* out,
                     int64_t ldc, int accumulate, void* stream);
int mhr_pack_tiles_t(const void* x, int64_t n_k, int64_t ld, int64_t n_sel, int tiles_per_block, void* out, void* stream);
int mhr_nce_wide_grad_tile(const void* queries_packed, int n_rows, const void* negs_packed, int n_neg, int dim, const uint16_t* bits,
                           const float* lse, const float* w, const float* scale_dev, const int32_t* n_live_dev, void* g_bf16,
                           int64_t ldg, void* stream);

/* Deterministic mode.  By default a few reductions of the loss backward add with float atomics from workgroups that arrive in any
 * order (the negative-side gradient tiles of mhr_nce_bwd_negs, the suppressed-pair corrections and d(logit_scale) of
 * mhr_nce_shared_bwd_rows, the per-offset loss sums of mhr_nce_finalize, the column sums of mhr_sum_rows_*): two runs of the same
 * step differ in the last bits.  mhr_set_deterministic(1) (process-wide, host side) makes the launchers that need no extra buffer
 * pick an order-independent form (mhr_sum_rows_*: one row range per column block; mhr_nce_finalize: one workgroup per group with
 * a fixed-order fold), and the two entry points with the optional arguments below then give one value whatever the arrival order:
 *   dn_fix    int64 [n_groups, n_neg, dim], zeroed: fixed-point (2^-40) accumulators that take what would be added to d_negs
 *             atomically; mhr_det_flush(dn_fix, d_negs, n) adds them into d_negs and zeroes them again;
 *   dq_fix / dp_fix (mhr_nce_bwd_tokens)  int64 shadows of dq_rows / dp_rows, zeroed; flushed the same way;
 *   dls_part  float, zeroed: one partial of d(logit_scale) per workgroup ([n_groups, 1024], mhr_nce_shared_bwd_rows) or per wave
 *             ([n_groups, 2048, 4], mhr_nce_bwd_tokens), without the factor exp(param);
 *             mhr_det_sum_into(dls_part, n, logit_scale_dev, 1, d_logit_scale) folds them in index order and applies
 *             exp(clamp(param, 0, ln 100)).
 * Same mathematics, one rounding more per accumulated value; bitwise reproducible run to run (and replayed vs host-issued). */
int mhr_set_deterministic(int on);
int mhr_get_deterministic(void);
int mhr_det_flush(int64_t* acc, float* dst, int64_t n, void* stream);
int mhr_det_sum_into(const float* parts, int64_t n, const float* scale_dev, int exp_clamped_scale, float* dst, void* stream);

/* Workspace queries (host functions, no launch).  The library allocates nothing: every entry point takes its operands, outputs
 * and scratch as caller-owned buffers whose shapes are stated with the declaration.  The only scratch whose SIZE depends on a
 * choice made inside the library is the scorers' candidate lists (how the item tiles are split into slices); these return the
 * total bytes of (cand_val f32, cand_idx i32, cand_cnt i32) for a launch - lay them out in that order:
 *   mhr_catalog_emit_slices(n_rows, n_sel_items)            n_slices to pass to mhr_catalog_score_emit_sliced
 *                                                           (n_sel_items = items the pass scores: ceil((n_items - begin) / stride))
 *   ..._emit_sliced_workspace_bytes(n_rows, n_sel_items, cap_s)   2 n_slices lists of cap_s slots per row
 *   ..._emit_wide_workspace_bytes(n_rows, cap_s)                  4 mhr_catalog_wide_slices(n_rows) lists of cap_s slots per row
 *   ..._emit_workspace_bytes(n_rows, cap)                         one list of cap slots per row
 *   ..._rows_dense_workspace_bytes(n_list, n_items)               one slot per (listed row, item) */
int mhr_catalog_emit_slices(int n_rows, int64_t n_sel_items);
int64_t mhr_catalog_score_emit_sliced_workspace_bytes(int n_rows, int64_t n_sel_items, int cap_s);
int64_t mhr_catalog_score_emit_wide_workspace_bytes(int n_rows, int cap_s);
int64_t mhr_catalog_score_emit_workspace_bytes(int n_rows, int cap);
int64_t mhr_catalog_score_rows_dense_workspace_bytes(int n_list, int64_t n_items);
/* Rows {row_begin + j * row_stride, j < n_sel} of x [n_rows, dim] bf16 (dim % 64 == 0) -> packed tile images
 * [ceil(n_sel / (32 tiles_per_block))][dim / 64][tiles_per_block][4096 B] (32 rows x 64 features each, XOR-swizzled like
 * the LDS tiles; rows past n_sel / n_rows are zero).  out: mhr_pack_tiles_bytes(n_sel, dim, tiles_per_block) bytes. */
int mhr_pack_tiles(const void* x, int64_t n_rows, int dim, int64_t row_begin, int64_t row_stride, int64_t n_sel,
                   int tiles_per_block, void* out, void* stream);
int64_t mhr_pack_tiles_bytes(int64_t n_sel, int dim, int tiles_per_block);
int mhr_catalog_score_emit_wide(const void* users, int n_rows, const void* items, int64_t n_items, int dim,
                                int64_t item_begin, int64_t item_stride, const uint32_t* tag_bits, const uint32_t* row_bits,
                                const float* tau, float* cand_val, int32_t* cand_idx, int32_t* cand_cnt, int n_slices,
                                int cap_s, void* stream);

/* Catalog masks on a dense score chunk (model/IDNet/hstu.py:982-999, trainer.py:724): column j is item
 * item_begin + j * item_stride; scores[r, j] = -inf unless (tag_bits[item] & row_bits[r]) != 0 and item != 0
 * (tag_bits NULL: every item carries bit 31 only).  In place. */
int mhr_catalog_mask_dense(float* scores, int64_t ld, int n_cols, int item_begin, int item_stride, const int32_t* tag_bits,
                           const int32_t* row_bits, int n_rows, void* stream);
/* Threshold emit from a dense score chunk into the list format of mhr_catalog_score_emit_sliced: the chunk's columns
 * (items item_begin + j) are cut into ceil(n_cols / seg) segments; segment i of row r appends its admissible scores
 * >= tau[r] to list list_base + i of [n_rows, n_lists, cap_s] (cand_cnt holds the unclamped count: > cap_s = overflow,
 * flagged by mhr_topk_select_sliced).  Append order within a list is arbitrary. */
int mhr_catalog_emit_dense(const float* scores, int64_t ld, int n_cols, int seg, int item_begin, const int32_t* tag_bits,
                           const int32_t* row_bits, const float* tau, int n_rows, float* cand_val, int32_t* cand_idx,
                           int32_t* cand_cnt, int n_lists, int list_base, int cap_s, void* stream);

/* Cross-head merge (collector.py:249-275): per user, the H*k per-head candidates are ordered by value
 * descending (ties: head, then rank ascending), first occurrences kept, first k returned with their
 * source head.  vals/idx [B, H, k] -> out_idx [B,k] i64, out_val [B,k] f32, out_src [B,k] i32.
 * status[b] = number of unique items found (k when the row is complete). */
int mhr_multihead_merge_dedup(const float* vals, const int64_t* idx, int B, int H, int k,
                              int64_t* out_idx, float* out_val, int32_t* out_src, int32_t* status, void* stream);

/* Hit matrix (collector.py:300-316): hit[b,j] = 1 when topk_idx[b,j] is among positives[b, 0:n_pos]. */
int mhr_hit_matrix(const int64_t* topk_idx, int B, int k, const int64_t* positives, int pos_stride, int n_pos,
                   uint8_t* hit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MHR_H */
