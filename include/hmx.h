/*
 * hmx.h -- C ABI of libhmx.so: the MI355X (gfx950) kernels behind the Hydra
 * generative-recommender hot path.
 *
 * The upstream project is pure PyTorch: it has no FFI, so the "interface each
 * entry point replaces" is the ATen call (or short chain of calls) at the cited
 * reference file:line.  The Python drop-in modules bind these symbols with ctypes.
 *
 * Conventions (all entry points):
 *   - plain device pointers and int64 sizes; `stream` is a hipStream_t passed as void*
 *     (NULL = the legacy default stream); work is enqueued, never synchronised;
 *   - no allocation: scratch comes in through (workspace, workspace_bytes), sized by the
 *     matching *_workspace_bytes() query, and is caller-owned;
 *   - returns HMX_OK (0) or a negative HMX_E* code; nothing is enqueued on error;
 *   - row-major contiguous tensors unless a stride argument says otherwise;
 *   - ids are int64 (torch.long).  An id outside [0, n_rows) is treated as the padding
 *     row (gathers zeros, receives no gradient) instead of faulting the GPU; the Python
 *     layer offers the reference's IndexError check as an opt-in host-side validation.
 */
#ifndef HMX_H
#define HMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMX_OK 0
#define HMX_EINVAL (-1)     /* bad shape / null pointer / unsupported size            */
#define HMX_EWORKSPACE (-2) /* workspace too small                                    */
#define HMX_ELAUNCH (-3)    /* HIP reported a launch error                            */
#define HMX_ELIMIT (-4)     /* input larger than a fixed-size entry point supports    */

int hmx_abi_version(void);
const char* hmx_error_string(int code);

/* ---------------------------------------------------------------------------------
 * Row gather.   out[i, :] = table[ids[i], :]
 * Replaces nn.Embedding.__call__ on item_emb / user_emb / pos_emb / action_emb /
 * time_span_emb / dwell_time_emb (embedding.py:157-164, :454, :630; hydra3.py:661-666)
 * and `item_table[sampled_hard_ids]` (hydra3.py:917).  Bit-exact copy.
 * ------------------------------------------------------------------------------- */
int hmx_gather_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n, float* out,
                        void* stream);

/* flags[r] = 1 when row r of x (n_rows, d) holds any value != 0 (NaN counts as non-zero), else 0.  The
 * data-parallel train step uses it on the dense table gradients (embedding.py:436-454 backward) to find the rows a
 * rank touched, so that only those rows are exchanged; no counterpart in the single-process reference. */
int hmx_row_any_nonzero_f32(const float* x, int64_t n_rows, int64_t d, uint8_t* flags, void* stream);

/* The same gather with the copy rounded to bf16 (round to nearest even) as it is written: out[i, :] =
 * bf16(table[ids[i], :]), for consumers that feed the rows to a bf16 MFMA product anyway (scoring tables, the
 * single-bf16 projection mode).  The write stream is half as wide.  d % 4 == 0; no backward (use the f32 form under
 * autograd).  Same id handling as hmx_gather_rows_f32. */
int hmx_gather_rows_bf16(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n, uint16_t* out,
                         void* stream);

/* Token-type routed gather (embedding.py:436-450):
 *   out[i] = item_table[ids[i]] if token_type[i]==1, user_table[ids[i]] if ==2, else 0. */
int hmx_gather_routed_f32(const float* item_table, int64_t item_rows, const float* user_table, int64_t user_rows,
                          int64_t d, const int64_t* ids, const int64_t* token_type, int64_t n, float* out,
                          void* stream);

/* out[i, :] = GELU(table[ids[i], :]) (exact erf form), the candidate-embedding generation of
 * embedding.py:654-735 in the ids-only configuration: get_complete_embeddings(..., fusion_style='late') is
 * F.gelu(item_emb[ids] + 0) (embedding.py:641-644).  Same id handling as hmx_gather_rows_f32 (an id out of
 * range reads as a zero row, GELU(0) = 0).  Inference only (no backward). */
int hmx_gather_gelu_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n, float* out,
                             void* stream);

/* ---------------------------------------------------------------------------------
 * Scatter-add of gradient rows (backward of the gathers above).
 *   grad_table[ids[i], :] += grad_rows[i, :]   for every i with ids[i] != padding_idx,
 *   ids[i] in range, and (token_type == NULL or token_type[i] == want_type).
 * Replaces ATen embedding_dense_backward / index_put_(accumulate=True) reached from
 * main.py:708 through embedding.py:445,450,454 and hydra3.py:917.
 * Deterministic, no float atomics: the (id, i) pairs are sorted stably by id; the sorted array is cut
 * at every change of id and at every multiple of 512 entries; each piece is summed in increasing i from
 * zero, the pieces of one id are added in order, and the total is added to the row once.  (For ids with
 * at most 512 occurrences not straddling a cut this is the plain sum in increasing i.)
 * grad_table is NOT cleared here (the caller zero-fills or accumulates).
 * row_flags (optional, n_rows bytes, NULL = not wanted): row_flags[r] = 1 is stored for every row r that receives a
 * sum (other entries are left alone).  A step touches a few hundred thousand of the 4 M rows of the dense gradient;
 * with the flags its consumers -- hmx_adamw_tables_step_f32, the data-parallel exchange -- read only those rows.
 * ------------------------------------------------------------------------------- */
size_t hmx_scatter_add_rows_workspace_bytes(int64_t n, int64_t n_rows, int64_t d);
int hmx_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                             int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                             int64_t padding_idx, uint8_t* row_flags, void* workspace, size_t workspace_bytes,
                             void* stream);

/* The same scatter-add for callers that cannot allocate device memory or hand over a workspace: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * The sort keys, their sorted copies, the run pieces and the sort's digit counters live in a fixed array of
 * hmx_scatter_add_rows_static_scratch_bytes() (16 MiB) inside the library, one copy per GPU, held until the process
 * ends; the call allocates nothing.  Inputs whose hmx_scatter_add_rows_workspace_bytes(n, n_rows, d) exceeds that
 * size (about 880,000 ids at d = 64) are refused with HMX_ELIMIT and nothing is enqueued.  Result: bit-identical to
 * hmx_scatter_add_rows_f32.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU
 * (different streams or threads) race; calls on one stream, or on different GPUs, do not.
 * The first call on a GPU looks up the array's address and raises two kernels' dynamic-LDS limits (HIP module
 * calls that are not allowed under stream capture): make one call per GPU (any size, n = 1 will do) before
 * capturing a graph that contains this entry point. */
size_t hmx_scatter_add_rows_static_scratch_bytes(void);
int hmx_scatter_add_rows_f32_static(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                                    int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                                    int64_t padding_idx, void* stream);

/* ---------------------------------------------------------------------------------
 * Full-table inner-product scoring fused with top-K (no score matrix in HBM).
 * Replaces torch.mm + masked_fill(-inf) + torch.topk of HardNegativeMiner
 * (hydra3.py:993-1006) and the external FAISS inner-product top-10 that infer.py:405-417
 * shells out to (exact instead of approximate).
 *
 * Score definition (what "exact" means, also used by the CPU oracle):
 *   s(q, j) = fma(q[d-1], t[j][d-1], ... fma(q[1], t[j][1], fma(q[0], t[j][0], 0)))  in fp32.
 * Result order / tie-break: score descending, then id ascending.
 *
 * hmx_table_prepare_bf16 converts a table shard once per weight update: bf16 copy for the
 * MFMA coarse pass plus the largest row 2-norm (for the coarse pass's error bound).
 * ------------------------------------------------------------------------------- */
int hmx_table_prepare_bf16(const float* table, int64_t n_rows, int64_t d, uint16_t* table_bf16, float* max_row_norm,
                           void* stream);

/* Exact path: fp32 MFMA scoring (bit-identical to the fma chain above) with a fused
 * per-query top-K.  exclude (optional, may be NULL): n_q x n_excl int64 ids (global ids) whose
 * scores count as -inf.  id_offset: global id of table row 0 (item-range shards).
 * Outputs: out_ids int64 (n_q, K) global ids, out_scores fp32 (n_q, K); slots beyond the
 * number of eligible items hold id -1 / score -inf. */
size_t hmx_score_topk_workspace_bytes(int64_t n_q, int64_t n_rows, int64_t d, int64_t k);
int hmx_score_topk_f32(const float* queries, int64_t n_q, const float* table, int64_t n_rows, int64_t d,
                       int64_t k, const int64_t* exclude, int64_t n_excl, int64_t id_offset, int64_t* out_ids,
                       float* out_scores, void* workspace, size_t workspace_bytes, void* stream);

/* The exact path for the queries with flags[q] != 0 only (flags: int32 per query on the device, e.g. the `uncertain`
 * output of hmx_score_topk_bf16): their rows of out_ids / out_scores are overwritten, every other row is left as it
 * is, and query tiles without a flagged query cost a few instructions.  Lets a caller repair the uncertain queries of
 * the fast path without reading their count back to the host. */
int hmx_score_topk_f32_flagged(const float* queries, int64_t n_q, const float* table, int64_t n_rows, int64_t d,
                               int64_t k, const int64_t* exclude, int64_t n_excl, int64_t id_offset, const int32_t* flags,
                               int64_t* out_ids, float* out_scores, void* workspace, size_t workspace_bytes, void* stream);

/* Fast path: bf16 MFMA coarse pass keeps a per-query shortlist, which is then re-scored
 * exactly (fma chain over the fp32 table) and ordered with the same tie-break.  A query
 * whose shortlist cannot be proven to contain the exact top-K (coarse-pass error bound vs
 * the shortlist's cut-off) is flagged in `uncertain` (int32 per query, 0/1); the caller
 * re-runs those queries through hmx_score_topk_f32.  Results for unflagged queries are
 * identical to hmx_score_topk_f32. */
int hmx_score_topk_bf16(const float* queries, int64_t n_q, const float* table, const uint16_t* table_bf16,
                        const float* max_row_norm, int64_t n_rows, int64_t d, int64_t k, const int64_t* exclude,
                        int64_t n_excl, int64_t id_offset, int64_t* out_ids, float* out_scores,
                        int32_t* uncertain, void* workspace, size_t workspace_bytes, void* stream);

/* Item-range shards that score the same queries can start from one another's sampled thresholds (no counterpart in
 * the reference, which has no multi-GPU path; the exact form it stands for is still hydra3.py:993 + :1006).
 * hmx_topk_sample_thresholds_bf16 writes, per query, the start threshold that hmx_score_topk_bf16 would sample from
 * this range (>= k sampled rows of the range score at least that much; -inf for launch shapes that do not sample);
 * n_ranges >= 1 is the number of ranges whose thresholds the caller will combine (each then samples its share of
 * what one table of the combined length would).  n_values = 1: thresholds is (n_q), the threshold itself.
 * n_values = 2 m (m <= 5 for k <= 10, <= 8 otherwise): thresholds is (n_q, 2 m), the m largest sampled scores of
 * each half of the sampled rows (distinct rows, descending per half); the k-th largest of the values of ALL ranges
 * is a tighter valid start than the largest per-range threshold.
 * The largest such value over the ranges is a valid start for every range, because the merged top-k only needs rows
 * that beat k rows of ANY range.  hmx_score_topk_bf16_seeded then runs the pass from start_thresholds (n_q) and
 * leaves the completeness decision to the caller, who sees all ranges: a range's list may hold fewer than k rows
 * (ids -1, scores -inf), bounds[q] is the score below which rows of this range may have been dropped (-inf: none
 * were), and the merged result of a query is complete iff its k-th merged score > the largest bounds[q] of the
 * ranges.  failed[q] != 0 only when the launch itself gave up (shared-stream hand-off); such queries, and those that
 * fail the test above, are re-run with hmx_score_topk_f32 on every range.  Same workspace size as
 * hmx_score_topk_bf16; no exclusion lists. */
int hmx_topk_sample_thresholds_bf16(const float* queries, int64_t n_q, const uint16_t* table_bf16, int64_t n_rows,
                                    int64_t d, int64_t k, int64_t n_ranges, int64_t n_values, float* thresholds,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* out[q] = k-th largest (k <= 16) of the n_slabs * n_values numbers vals[s * slab_stride + q * n_values + v], -inf when
 * there are fewer than k: the combine step for the sampled scores of several ranges, read as all_gather leaves them
 * (one slab per range). */
int hmx_kth_largest_f32(const float* vals, int64_t n_slabs, int64_t slab_stride, int64_t n_q, int64_t n_values, int64_t k,
                        float* out, void* stream);
int hmx_score_topk_bf16_seeded(const float* queries, int64_t n_q, const float* table, const uint16_t* table_bf16,
                               const float* max_row_norm, int64_t n_rows, int64_t d, int64_t k, int64_t id_offset,
                               const float* start_thresholds, int64_t* out_ids, float* out_scores, float* bounds,
                               int32_t* failed, void* workspace, size_t workspace_bytes, void* stream);

/* Test hook (no counterpart in the reference).  The shared-stream form of the bf16 coarse pass hands item rows from
 * wave to wave through counters in LDS with bounded waits; a wait that runs out makes the whole workgroup stop
 * early and flags every query of the launch `uncertain`.  This call arms a one-shot fault for the NEXT launch of
 * that kernel in this process: workgroup 0 leaves out the publication of stage `stage` (>= 0; negative disarms), so
 * its waves run into the bound.  Used by the GPU test of that path; not thread-safe, never needed in production. */
int hmx_debug_ring_skip_publish(int64_t stage);

/* Merge S per-shard top-K lists (cand_* laid out (S, n_q, K), as RCCL all_gather leaves
 * them) into the global top-K with the same order / tie-break.  Replaces nothing in the
 * reference (it is single-device); it is the exchange step of item-range sharding. */
int hmx_topk_merge(const int64_t* cand_ids, const float* cand_scores, int64_t n_shards, int64_t n_q, int64_t k,
                   int64_t* out_ids, float* out_scores, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused InfoNCE loss (compute_mixed_contrastive_loss, hydra3.py:845-974).
 * For every query row i of q (m, d):
 *   logit[i, 0]                = <q_i^, k_i^> / temperature                     (positive, hydra3.py:958)
 *   logit[i, 1 .. n_hard]      = <q_i^, table[hard_ids[i, .]]^> / temperature   (hard negatives gathered from the
 *                                                                                item table, hydra3.py:917-919)
 *   logit[i, 1 + n_hard .. ]   = <q_i^, k[inb_idx[i, .]]^> / temperature        (in-batch negatives, hydra3.py:925-950)
 *   row_loss[i]                = logsumexp(logit[i, :]) - logit[i, 0]            (hydra3.py:965-967)
 * x^ = x / max(|x|, 1e-12) when normalize != 0 (F.normalize), else x.  An id / index out of range is a zero key.
 * The caller draws hard_ids and inb_idx (any sampling policy) and takes the mean of row_loss.
 * Forward leaves logits (m, 1 + n_hard + n_inb), lse (m), q_inv (m), k_inv (m) for the backward pass.
 * dq_unit (m, d), optional (NULL: not wanted): the forward's one walk over the keys also accumulates
 * d row_loss[i] / d q[i] (soft-max weighted key sum on a running maximum, then the normalisation Jacobian) and leaves
 * it there; a backward call that is given the same buffer forms dq = grad_row_loss[i] * dq_unit[i] and the key
 * coefficients from logits and lse without walking the keys a second time.  With NULL in both calls the backward
 * walks the keys again (same values up to fp32 summation order).
 * Backward takes grad_row_loss (m) and produces dq (m, d), dk (m, d) (both overwritten) and, when n_hard > 0,
 * ADDS the hard-negative gradient rows into dtable (n_rows, d), which must be zero on entry (the normalisation
 * Jacobian is applied in place to the rows it touched); dtable_flags (optional): see row_flags of
 * hmx_scatter_add_rows_f32.  `logits` is overwritten.  d % 4 == 0, d <= 256,
 * 16-byte aligned q / k / table / dq / dk / dtable.  Deterministic (sorted-run sums, no float atomics).
 * Index arguments.  hard_ids is (m, hard_ld) with hard_ld = 0 meaning n_hard; a shorter row (hard_ld < n_hard) is
 * allowed where the slots beyond it repeat (the periodic / grouped calls, period <= hard_ld), so the tiled copy of
 * hydra3.py:911-914 need not exist.  inb_idx is int64 (m, n_inb) unless inb_flags has HMX_NCE_INB_I32 (then int32);
 * with HMX_NCE_INB_SKIP_SELF as well, the int32 values are the raw draws r of hydra3.py:824 in [0, m - 1) and row i's
 * key is row r + (r >= i) (hydra3.py:826-829), applied on the fly. */
#define HMX_NCE_INB_I32 1
#define HMX_NCE_INB_SKIP_SELF 2
int hmx_nce_fwd_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table, int64_t n_rows,
                    const int64_t* hard_ids, int64_t n_hard, const void* inb_idx, int64_t n_inb, int64_t hard_ld,
                    int64_t inb_flags, float temperature, int normalize, float* logits, float* row_loss, float* lse,
                    float* q_inv, float* k_inv, float* dq_unit, void* stream);
size_t hmx_nce_bwd_workspace_bytes(int64_t m, int64_t d, int64_t n_rows, int64_t n_hard, int64_t n_inb);
int hmx_nce_bwd_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table, int64_t n_rows,
                    const int64_t* hard_ids, int64_t n_hard, const void* inb_idx, int64_t n_inb, int64_t hard_ld,
                    int64_t inb_flags, float temperature, int normalize, float* logits, const float* lse, const float* q_inv,
                    const float* k_inv, const float* grad_row_loss, const float* dq_unit, float* dq, float* dk,
                    float* dtable, uint8_t* dtable_flags, void* workspace, size_t workspace_bytes, void* stream);
/* Hard negatives that repeat, which is how hydra3.py:905-919 builds them (each user's K mined ids, tiled over the n_hard
 * slots of all of that user's positions).
 * hmx_nce_fwd_periodic_f32: the caller guarantees hard_ids[i, s] == hard_ids[i, s % period] for every row; only the
 * first `period` hard slots are scored and each enters the log-sum-exp as many times as it is repeated.  Same row_loss
 * as hmx_nce_fwd_f32 up to fp32 summation order; of the hard columns of `logits` only the first `period` are written.
 * hmx_nce_bwd_grouped_f32: the backward for logits that came from hmx_nce_fwd_periodic_f32 with the same period, when
 * in addition the rows form groups [group_start[g], group_start[g + 1]) (n_groups + 1 ascending offsets into the m
 * rows, on the device) whose rows all carry the list of the group's first row.  The table gradient is then formed
 * per group as a (period x rows-of-the-group) x (rows x d) product and only n_groups * period rows are scattered
 * instead of m * n_hard.  dq, dk and the table rows equal hmx_nce_bwd_f32's up to fp32 summation order.
 * period <= 64, period <= n_hard. */
int hmx_nce_fwd_periodic_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table, int64_t n_rows,
                             const int64_t* hard_ids, int64_t n_hard, int64_t period, const void* inb_idx,
                             int64_t n_inb, int64_t hard_ld, int64_t inb_flags, float temperature, int normalize,
                             float* logits, float* row_loss, float* lse,
                             float* q_inv, float* k_inv, float* dq_unit, void* stream);
size_t hmx_nce_bwd_grouped_workspace_bytes(int64_t m, int64_t d, int64_t n_rows, int64_t n_hard, int64_t n_inb,
                                           int64_t n_groups, int64_t period);
int hmx_nce_bwd_grouped_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table, int64_t n_rows,
                            const int64_t* hard_ids, int64_t n_hard, const void* inb_idx, int64_t n_inb, int64_t hard_ld,
                            int64_t inb_flags, float temperature, int normalize, float* logits, const float* lse,
                            const float* q_inv,
                            const float* k_inv, const float* grad_row_loss, const float* dq_unit, float* dq, float* dk,
                            float* dtable, uint8_t* dtable_flags, const int64_t* group_start, int64_t n_groups,
                            int64_t period, void* workspace, size_t workspace_bytes, void* stream);

/* =================================================================================
 * Sequence encoder (hydra3.py).  Activations are (T = B*L tokens, features) row-major fp32.
 * Forward and backward of every stage; the Python modules chain them under autograd.
 * ================================================================================= */

/* RMSNorm, y = x / sqrt(mean(x^2) + eps) * scale (hydra3.py:221-229).  rstd (T) is kept for the
 * backward pass (may be NULL).  Backward writes dx and per-block partials of dscale, shaped
 * (hmx_rmsnorm_bwd_blocks(T), d); their column sums (hmx_colsum_f32) are dscale.  d <= 256. */
int hmx_rmsnorm_fwd(const float* x, const float* scale, int64_t T, int64_t d, float eps, float* y, float* rstd,
                    void* stream);
int64_t hmx_rmsnorm_bwd_blocks(int64_t T);
int hmx_rmsnorm_bwd(const float* x, const float* scale, const float* rstd, const float* dy, int64_t T, int64_t d,
                    float* dx, float* dscale_part, void* stream);

/* out[c] = sum_r P[r * ld + c], fixed summation order (reduction of per-block / per-sequence partials).
 * With a workspace of hmx_colsum_workspace_bytes(rows, cols) tall inputs are summed in row chunks by many
 * workgroups and the chunks added in order; workspace NULL: one workgroup per 64 columns. */
size_t hmx_colsum_workspace_bytes(int64_t rows, int64_t cols);
int hmx_colsum_f32(const float* P, int64_t rows, int64_t cols, int64_t ld, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Dense projection on bf16 MFMA with fp32 accumulate (nn.Linear / F.linear at hydra3.py:206,210,263,
 * 275,366,388,416-418,643 and their dgrad):
 *   Y[t, n] = ( sum_k X[t, k] * in_rowmask[t] * Wv(n, k) + bias[n] + c1 R1[t, n] + c2 R2[t, n] ) * out_rowmask[t]
 * with Wv(n, k) = W[n * w_rs + k * w_cs] (w_rs = K, w_cs = 1 for a forward product with an (N, K) weight;
 * w_rs = 1, w_cs = ld for the transposed product of the backward pass).  Any of bias, R1, R2, the masks
 * may be NULL.  accumulate != 0 adds the result to Y.  K <= 256.
 * Precision (process-wide, hmx_set_linear_precision): split_bf16 = 1 (default) writes every operand as a sum of
 * two bf16 numbers and accumulates hi*hi + lo*hi + hi*lo on the MFMA (fp32-class accuracy, three MFMAs per
 * product); split_bf16 = 0 rounds operands to one bf16 (relative error ~2^-9 per operand, one MFMA). */
int hmx_set_linear_precision(int split_bf16);
int hmx_linear_f32(const float* X, int64_t ldx, const float* W, int64_t w_rs, int64_t w_cs, const float* bias,
                   int64_t T, int64_t N, int64_t K, float* Y, int64_t ldy, const float* R1, int64_t ldr1, float c1,
                   const float* R2, int64_t ldr2, float c2, const float* in_rowmask, const float* out_rowmask,
                   int accumulate, void* stream);

/* The per-head projections of the multi-head Mamba blocks (hydra3.py:156 deltaBC_layer, 243/248 in_proj / out_proj of
 * every head in the hydra3.py:366 loop, and their input gradients) as ONE launch: head h < n_heads (<= 8) computes
 *   Y[t, h * y_head_stride + n] (+)= sum_k X[t, h * x_head_stride + k] * Wv_h(n, k),  Wv_h(n, k) = W[h][n * w_rs + k * w_cs]
 * with the precision of hmx_linear_f32.  W: HOST array of n_heads device pointers (the heads' weights are separate
 * parameters).  accumulate != 0 adds to Y.  Same results as n_heads calls of hmx_linear_f32 on the column blocks. */
int hmx_heads_linear_f32(const float* X, int64_t ldx, int64_t x_head_stride, const float* const* W, int64_t w_rs,
                         int64_t w_cs, int64_t T, int64_t N, int64_t K, float* Y, int64_t ldy, int64_t y_head_stride,
                         int n_heads, int accumulate, void* stream);

/* Weight / bias gradient (the dW / db of every nn.Linear above): dW[n * ldw + k] = sum_t dY[t, n] * X[t, k] * in_rowmask[t],
 * dbias[n] = sum_t dY[t, n] (dbias may be NULL).  Precision (process-wide, hmx_set_wgrad_precision): fp32_mfma = 0
 * (default) writes every operand as a sum of two bf16 numbers and accumulates hi*lo + lo*hi + hi*hi on the bf16 MFMA in
 * fp32 (16 significant bits per operand, as hmx_linear_f32 does); fp32_mfma = 1 multiplies on the fp32 MFMA (5 x
 * slower multiply rate).  The bias sum is fp32 either way.  A wave owns up to eight 32 x 32 output tiles and reads its operands
 * with vector loads where base, row stride and extent allow (16-byte aligned dY / X with ldy, ldx, N, K multiples of
 * four give the widest tiles; anything else still works, with more re-reads of the operands).
 * One launch: token splits write partial tiles to the workspace and
 * the workgroup that finishes a tile last adds them in split order (deterministic, no float atomics).  The arrival
 * counters of that hand-over live inside the library (zero between launches), one set per (GPU, stream): launches on
 * up to sixteen streams of a GPU may be in flight at once (a seventeenth stream takes over the set of the stream
 * that launched longest ago, whose launch must have completed by then). */
typedef struct hmx_wgrad_desc {
    const float* dY;          /* (T, N) with row stride ldy                                   */
    int64_t ldy;
    const float* X;           /* (T, K) with row stride ldx                                   */
    int64_t ldx;
    const float* in_rowmask;  /* (T) or NULL                                                  */
    int64_t T, N, K;
    float* dW;                /* (N, K) with row stride ldw                                   */
    int64_t ldw;
    float* dbias;             /* (N) or NULL                                                  */
    int32_t accumulate;       /* 0: dW / dbias are overwritten; 1: the result is added to them */
    /* Grouped outputs (all four 0: none).  dY's N columns are n_group_padded-wide groups of which the first n_group
     * are real: column g * n_group_padded + r (r < n_group) is output row g * n_group + r of dW / dbias, the other
     * columns are skipped; the same along K with the k_ fields.  This is how the widened activations of a fused
     * block address the true weight: N, K, ldy, ldx describe the widened operands, dW / ldw / dbias the true ones. */
    int32_t n_group, n_group_padded, k_group, k_group_padded;
} hmx_wgrad_desc;
int hmx_set_wgrad_precision(int fp32_mfma);
size_t hmx_linear_wgrad_workspace_bytes(int64_t T, int64_t N, int64_t K);
int hmx_linear_wgrad_f32(const float* dY, int64_t ldy, const float* X, int64_t ldx, const float* in_rowmask, int64_t T,
                         int64_t N, int64_t K, float* dW, int64_t ldw, float* dbias, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Up to 24 weight-gradient products in ONE launch (descs is a host array, read before the call returns).  The
 * backward pass of an encoder block has some thirty of these products, each a few microseconds of work: batched,
 * they fill the chip once instead of thirty times.  No two descriptors of a call may write the same dW / dbias. */
size_t hmx_linear_wgrad_batch_workspace_bytes(const hmx_wgrad_desc* descs, int64_t n);
int hmx_linear_wgrad_batch_f32(const hmx_wgrad_desc* descs, int64_t n, void* workspace, size_t workspace_bytes,
                               void* stream);

/* s = SiLU(x1); y2 = RoPE(s) on interleaved pairs per head, position = t mod L, angle tables cos/sin of
 * shape (L, hd/2) (hydra3.py:300-357, 368, 381-386).  Backward: dx1 = (ds + RoPE^T(dy2)) * SiLU'(x1). */
int hmx_silu_rope_fwd(const float* x1, const float* cos_t, const float* sin_t, int64_t T, int64_t L, int64_t d,
                      int64_t hd, float* s, float* y2, void* stream);
int hmx_silu_rope_bwd(const float* x1, const float* cos_t, const float* sin_t, const float* ds, const float* dy2,
                      int64_t T, int64_t L, int64_t d, int64_t hd, float* dx1, void* stream);

/* mode 0: out = a * b * c (y1 * y2 / sqrt(H), hydra3.py:387);  mode 1: out = SiLU(a) * b (hydra3.py:416-418). */
int hmx_binary_fwd(const float* a, const float* b, int64_t n, float c, int mode, float* out, void* stream);
int hmx_binary_bwd(const float* a, const float* b, const float* g, int64_t n, float c, int mode, float* da, float* db,
                   void* stream);

/* out[t, :] = (c0 x0 + c1 x1 + c2 x2 + c3 x3)[t, :] * rowmask[t]; x1..x3 and rowmask may be NULL
 * (the sums and mask of hydra3.py:647-669 and the mask / coefficient steps of the backward pass). */
int hmx_combine_f32(const float* x0, const float* x1, const float* x2, const float* x3, float c0, float c1, float c2,
                    float c3, const float* rowmask, int64_t T, int64_t d, float* out, void* stream);

/* Index preparation of log2feats (hydra3.py:621-659): mask[t] = (id != 0), pos_idx = (l + 1) * (id != 0),
 * time_bin = min(floor(ln(dt + 1)), time_span - 1) * (id != 0), dt = ts[t] - ts[t-1] unless t starts a
 * sequence or follows a padding token. */
int hmx_seq_index(const int64_t* ids, const int64_t* ts, int64_t T, int64_t L, int64_t time_span, float* mask,
                  int64_t* pos_idx, int64_t* time_bin, void* stream);

/* Causal depthwise conv (kernel 4) + bias + SiLU over x (T, C), weights w (C, 4), b (C)
 * (nn.Conv1d at hydra3.py:243-250 with the slice at :267, SiLU at :269).  Backward writes dx and
 * per-sequence partials part (B, 5 C) = [dw (C, 4) | db (C)]; their sums over B are the parameter gradients. */
int hmx_conv_silu_fwd(const float* x, const float* w, const float* b, int64_t T, int64_t L, int64_t C, float* xc,
                      void* stream);
int hmx_conv_silu_bwd(const float* x, const float* w, const float* b, const float* dxc, int64_t B, int64_t L,
                      int64_t C, float* dx, float* part, void* stream);

/* One-query multi-head attention of the multi-scale fusion tail (hydra3.py:528-533: nn.MultiheadAttention with
 * query = the short-term encoder's last token, key = value = the long-term representation; no key-padding mask).
 * q (B, d) and kv (B * L, 2d) = [k | v] are the ALREADY PROJECTED inputs (in_proj through hmx_linear_f32), heads are
 * column blocks of d / H.  out[b, h-block] = sum_l softmax_l(<q, k_l> / sqrt(d/H)) * drop[b, h, l] * v_l.
 * drop (B, H, L) is the dropout multiplier of the attention weights (NULL in eval).  p (B, H, L) receives the
 * softmax weights for the backward pass, which writes dq (B, d) and dkv (B * L, 2d).  L <= 4096. */
int hmx_attn1q_fwd(const float* q, const float* kv, const float* drop, int64_t B, int64_t L, int64_t d, int64_t H,
                   float* p, float* out, void* stream);
int hmx_attn1q_bwd(const float* q, const float* kv, const float* drop, const float* p, const float* dout, int64_t B,
                   int64_t L, int64_t d, int64_t H, float* dq, float* dkv, void* stream);

/* pscan(A, X) (hydra3.py:16-104): H[b, t] = A[b, t] * H[b, t-1] + X[b, t], H[b, -1] = 0, for A, X, H of shape
 * (B, L, C) with C = ED * N flattened.  The reference evaluates it with a Blelloch up/down sweep over L; this is
 * the same recurrence evaluated sequentially in t (one chain per thread), so results agree to fp32 rounding.
 * Backward: dX[t] = g[t], dA[t] = g[t] * H[t-1] with g[t] = dH[t] + A[t+1] * g[t+1]. */
int hmx_pscan_fwd(const float* A, const float* X, int64_t B, int64_t L, int64_t C, float* H, void* stream);
int hmx_pscan_bwd(const float* A, const float* H, const float* dH, int64_t B, int64_t L, int64_t C, float* dA,
                  float* dX, void* stream);

/* Selective scan of all heads (hydra3.py:107-123, 203-214) fused with the dt projection + softplus
 * (:210) and the SiLU(z) gate (:273).  C = H * ED channels, ED a power of two >= 32, state N = 16,
 * dt rank 16.  dbc (T, H, 48) = [dt_rank | B | C] rows of deltaBC_layer.  hs (hmx_scan_state_floats(B, L, C)
 * floats), if not NULL, receives state checkpoints for the backward pass (one every four steps, in the kernels' own
 * order: opaque to callers, to be handed to hmx_scan_bwd unchanged; the backward pass recomputes the states in
 * between instead of reading them back).  Backward writes dxc, dz, per-group partials of d(dbc)
 * shaped (T, H, hmx_scan_groups_per_head(ED), 48) (sum with hmx_sum_groups_f32) and per-sequence
 * parameter partials par_part (B, 34 C) = [dA_log (C, 16) | dW_dt (C, 16) | db_dt (C) | dD (C)]. */
int64_t hmx_scan_state_floats(int64_t B, int64_t L, int64_t C);
int hmx_scan_fwd(const float* xc, const float* z, const float* dbc, const float* Wdt, const float* bdt,
                 const float* A_log, const float* D, int64_t B, int64_t L, int64_t C, int64_t ED, int64_t H, float* yg,
                 float* hs, void* stream);
int64_t hmx_scan_groups_per_head(int64_t ED);
int hmx_scan_bwd(const float* xc, const float* z, const float* dbc, const float* Wdt, const float* bdt,
                 const float* A_log, const float* D, const float* hs, const float* dyg, int64_t B, int64_t L, int64_t C,
                 int64_t ED, int64_t H, float* dxc, float* dz, float* ddbc_part, float* par_part, void* stream);
int hmx_sum_groups_f32(const float* part, int64_t rows, int64_t groups, int64_t width, float* out, void* stream);

/* -------------------------------------------------------------------------------
 * Input assembly of the sequence encoder in one pass (hydra3.py:621-669 with embedding.py:436-450, 613-652 for the
 * ids-plus-positions half of the early-fusion embedding):
 *   pm[t]  = ids[t] != 0;  pos[t] = pm ? (t mod L) + 1 : 0;  time_bin[t] per hydra3.py:650-659
 *   e[t]   = (token_type 1: item_table[ids[t]], 2: user_table[ids[t]], else 0) + pos_table[pos[t]]
 *   out[t] = ((W[:, :ds] e[t] + bias + pre[t]) * scale + time_rows[time_bin[t]] + action_rows[action[t]]
 *             + dwell_rows[dwell[t]]) * pm[t]
 * W is hydra_projection's (d, ldw) weight whose first ds columns meet the id embedding; `pre` (T, d) or NULL carries
 * what the other columns contribute (the feature half, zero in an ids-only batch); time / action / dwell rows are the
 * three small tables after embedding_projection (the tables themselves when that is the identity).  The projection runs
 * on bf16 MFMA with split operands (fp32-class, like hmx_linear_f32); gathered rows are never written unless `e` is
 * given (training: the X operand of the weight gradient).  Out-of-range action / dwell / time ids read as zero rows.
 * Backward: g = dout * pm, gs = g * scale, de = W[:, :ds]^T gs; the caller scatter-adds de into the item / user /
 * position tables and g into the small tables (hmx_scatter_add_rows_f32) and queues (gs, e) for the weight gradient.
 * Shapes: d and ds multiples of 32, d <= 128, ds <= 256 (hmx_input_assembly_supported); pos_rows >= L + 1; tables,
 * rows and outputs 16-byte aligned. */
typedef struct hmx_input_assembly_args {
    int64_t T, L, ds, d, time_span;
    const int64_t *ids, *token_type, *timestamps, *action, *dwell;      /* (T) each                          */
    const float *item_table, *user_table, *pos_table;                   /* (rows, ds)                        */
    int64_t item_rows, user_rows, pos_rows;
    const float* W;                                                     /* (d, ldw)                          */
    int64_t ldw;
    const float* bias;                                                  /* (d)                               */
    const float* pre;                                                   /* (T, d) or NULL                    */
    float scale;
    const float *time_rows, *action_rows, *dwell_rows;                  /* (n, d) each                       */
    int64_t time_n, action_n, dwell_n;
    float* out;                                                         /* (T, d)                            */
    float* mask;                                                        /* (T): pm, written by fwd, read by bwd */
    int64_t *pos_idx, *time_bin;                                        /* (T) each, written by fwd          */
    float* e;                                                           /* (T, ds) or NULL, written by fwd   */
    const float* dout;                                                  /* backward: (T, d)                  */
    float *g, *gs;                                                      /* backward: (T, d) each             */
    float* de;                                                          /* backward: (T, ds)                 */
} hmx_input_assembly_args;
int hmx_input_assembly_supported(int64_t d, int64_t ds);
int hmx_input_assembly_fwd(const hmx_input_assembly_args* a, void* stream);
int hmx_input_assembly_bwd(const hmx_input_assembly_args* a, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused halves of a Hydra block (HydraBlockFirst / HydraBlock, hydra3.py:392-457, with MultiHeadMamba :359-389).
 * Everything of a block that has no dependence between tokens runs in two kernels per direction:
 *   pre  half: dropout multiplier, RMSNorm1, linear_layer1 (+ bias, input mask), SiLU, RoPE branch (y2), in_proj of
 *              every head (x_in, z)                                   (hydra3.py:411-412 / 445-446, 366-386, 263-265)
 *   post half: out_proj of every head, y1 * y2 / sqrt(H), linear_layer2 + bias + x_org + block residual, RMSNorm2,
 *              down(SiLU(gate) * up) + residual, output mask          (hydra3.py:275, 387-388, 413-423 / 447-457)
 * The causal conv, deltaBC projection and selective scan between the halves stay sequence kernels.
 * Weights are packed once per forward pass (hmx_block_pack_f32: bf16 hi / lo fragments, both orientations) into a
 * buffer of hmx_block_packed_bytes(d, H).  Supported shapes: hmx_block_supported(d, H) (d and d / H multiples of 32,
 * d <= 128, H <= 8); other shapes use the per-operation entry points above.
 * The backward halves recompute the intermediates they need from (x, m, rstd2, yg, y2), write the input gradients
 * (dyg, dy2, dm; then dx) and the operand pairs of every weight gradient (hmid, n2, dg, du, p, dy1; s, dx1, n) for
 * hmx_linear_wgrad_batch_f32, plus per-wave partials of the two RMSNorm scale gradients, shaped
 * (hmx_block_scale_partial_rows(T), d) (sum with hmx_colsum_f32; a launch writes every row of that buffer -- the
 * rows beyond its waves as zeros -- so the buffer need not be cleared).
 * One argument block serves the four calls; each reads the fields its half needs.
 * Widened blocks.  A width whose head_dim is not a multiple of 32 -- the reference's default hidden size 60 gives
 * head_dim 30 (main.py:54, :57) -- runs on the same kernels in a widened layout: every head's head_dim features are
 * followed by zero features up to the next multiple of 32, every head's 4 * head_dim inner channels by zero channels
 * up to 4 times that (60 -> 64, 240 -> 256).  hmx_block_padded_width(d_real, H) is the width d to pass everywhere
 * (0: not supported; d_real itself: nothing to widen); hmx_block_pack_f32 takes the TRUE weights (shapes in d_real)
 * and writes images with zero rows / columns at the added positions, plus widened copies of the four d-vectors, which
 * the kernels then read instead of the scale1 / b1 / b2 / scale2 arguments; every activation argument is in the
 * widened layout (the caller zero-fills the added features of x once; they stay zero through every block), and the
 * RMSNorm means run over d_real.  Weight-gradient operands come out widened as well: the caller's
 * hmx_wgrad_desc entries address the true sub-blocks (per head pair) of each product.
 * ------------------------------------------------------------------------------- */
typedef struct hmx_block_weights {
    const float *W1, *W2, *Wg, *Wu, *Wd;   /* linear_layer1, linear_layer2, gate, up, down: (d, d)          */
    const float* Win[8];                   /* per head in_proj (2 * ED, hd), rows [x | z]                   */
    const float* Wout[8];                  /* per head out_proj (hd, ED)                                    */
    const float *scale1, *b1, *b2, *scale2; /* (d) each: norm1.scale, linear_layer1.bias, linear_layer2.bias,
                                              norm2.scale; read only when the block is packed widened        */
} hmx_block_weights;
typedef struct hmx_block_args {
    int64_t T, L, d, H;
    int32_t first;                         /* 1: HydraBlockFirst residual wiring, 0: HydraBlock             */
    float eps;
    const void* packed;
    const float *mask, *x, *drop, *scale1, *b1, *rope_cos, *rope_sin;   /* drop (T, d) multiplier or NULL   */
    float *rstd1, *x_in, *z, *y2;          /* pre forward outputs: (T), (T, H*ED), (T, H*ED), (T, d)        */
    const float *yg, *b2, *scale2;         /* post forward inputs: gated scan output (T, H*ED)              */
    float *out, *m, *rstd2;                /* post forward outputs: (T, d), (T, d), (T)                     */
    const float* dout;                     /* post backward input (T, d)                                    */
    float *dyg, *dy2, *dm, *hmid, *n2, *dg, *du, *p, *dy1, *dscale_part;
    const float *dx_in, *dz;               /* pre backward inputs (T, H*ED) each                            */
    float *dx, *s, *dx1, *n;
    int64_t d_real;                        /* true model width of a widened block (0 or d: not widened)     */
} hmx_block_args;
int hmx_block_supported(int64_t d, int64_t H);
int64_t hmx_block_padded_width(int64_t d, int64_t H);
size_t hmx_block_packed_bytes(int64_t d, int64_t H);
int64_t hmx_block_scale_partial_rows(int64_t T);
int hmx_block_pack_f32(const hmx_block_weights* w, int64_t d, int64_t H, int64_t d_real, void* packed, void* stream);
int hmx_block_pre_fwd(const hmx_block_args* a, void* stream);
int hmx_block_post_fwd(const hmx_block_args* a, void* stream);
int hmx_block_post_bwd(const hmx_block_args* a, void* stream);
int hmx_block_pre_bwd(const hmx_block_args* a, void* stream);

/* ---------------------------------------------------------------------------------
 * Benchmark support (no counterpart in the reference).  When enabled, entry points bracket
 * their dominant kernel with HIP events on the caller's stream; hmx_profile_collect blocks
 * until the events of spans tagged `tag` have completed, returns their summed duration and
 * count, and forgets them.  Tags: "topk_coarse_bf16", "topk_coarse_f32", "gather_rows", "gather_rows_bf16", "scatter_segment_sum".
 * ------------------------------------------------------------------------------- */
int hmx_profile_enable(int on);
/* Restricts recording to spans with this tag (NULL or "": every tag again).  Every recorded span costs two event
 * records on the stream; a timed region that wants one kernel's duration records that kernel alone. */
int hmx_profile_filter(const char* tag);
int hmx_profile_collect(const char* tag, double* total_ms, int64_t* launches);

/* =================================================================================
 * Optimizer end of the train step: clip_grad_norm_(max_norm) + torch.optim.AdamW.step() of main.py:640,709-713 over a
 * list of fp32 tensors.  `tensors` and `chunk_start` are DEVICE arrays: n_tensors descriptors, and n_tensors + 1
 * ascending int32 offsets with chunk_start[i + 1] - chunk_start[i] = ceil(tensors[i].n / hmx_adamw_chunk_elems()),
 * total_chunks = chunk_start[n_tensors].  n_tensors <= 1023.
 * hmx_grad_norm_f32: norm_out[0] = sqrt(sum of squares of all g, tensors and tables) (clip_grad_norm_'s total norm),
 *   norm_out[1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0); both stay on the device.  Deterministic:
 *   per-chunk partial sums added in a fixed order.
 * hmx_adamw_step_f32: for every element, with g' = g * (*grad_scale) (grad_scale: device scalar or NULL = 1):
 *   p *= 1 - lr * weight_decay;  m += (g' - m) * (1 - beta1);  v = beta2 * v + (1 - beta2) * g' * g';
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *   (decoupled weight decay, no amsgrad, not maximize: the reference's configuration).  step counts from 1.
 *   write_scaled_grads != 0 also stores g' back into g, which is what clip_grad_norm_ leaves in .grad.
 * ================================================================================= */
typedef struct {
    float* p;        /* parameter, updated in place */
    const float* g;  /* its gradient */
    float* m;        /* exp_avg */
    float* v;        /* exp_avg_sq */
    int64_t n;       /* elements */
} hmx_adamw_tensor;
/* An embedding table of the same update, walked by rows (`tables` arguments are HOST arrays of at most eight, read
 * before the call returns).  d % 4 == 0, d <= 256, 16-byte aligned p / g / m / v.
 * row_flags (n_rows bytes or NULL): as left by hmx_scatter_add_rows_f32 -- rows with flag 0 hold a zero gradient and
 *   it is not read; a flagged row's gradient is set back to zero and its flag cleared once the update has used it,
 *   so that after the step g is again the all-zero dense gradient the next backward pass adds into (the zero-fill
 *   of main.py:684's zero_grad and the reads of a gigabyte of zeros are what this saves).  NULL: every row is read
 *   and g is left as it is.
 * bf16_out (n_rows, bf16_ld) or NULL: receives the updated rows rounded to bf16 (the scoring copy that
 *   hmx_table_prepare_bf16 would make; columns d .. bf16_ld of a row are not written) and bf16_max_norm the largest
 *   updated row 2-norm. */
typedef struct hmx_adamw_table {
    float* p;
    float* g;
    float* m;
    float* v;
    int64_t n_rows, d;
    uint8_t* row_flags;
    uint16_t* bf16_out;
    int64_t bf16_ld;
    float* bf16_max_norm;
} hmx_adamw_table;
int64_t hmx_adamw_chunk_elems(void);
int64_t hmx_adamw_table_blocks(int64_t n_rows);
/* workspace for hmx_grad_norm_f32: n_partials = total_chunks + the sum of hmx_adamw_table_blocks(n_rows) of the tables */
size_t hmx_grad_norm_workspace_bytes(int64_t n_partials);
int hmx_grad_norm_f32(const hmx_adamw_tensor* tensors, const int32_t* chunk_start, int64_t n_tensors,
                      int64_t total_chunks, const hmx_adamw_table* tables, int64_t n_tables, float max_norm,
                      void* workspace, size_t workspace_bytes, float* norm_out, void* stream);
int hmx_adamw_tables_step_f32(const hmx_adamw_table* tables, int64_t n_tables, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream);
int hmx_adamw_step_f32(const hmx_adamw_tensor* tensors, const int32_t* chunk_start, int64_t n_tensors,
                       int64_t total_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t step, const float* grad_scale, int write_scaled_grads, void* stream);

/* ---------------------------------------------------------------------------------
 * Feature-fusion paths of the embedding module (embedding.py:216-264, 486-611).  Sparse features use
 * hmx_gather_rows_f32, every nn.Linear uses hmx_linear_f32; the calls below are the remaining row operations.
 * ------------------------------------------------------------------------------- */

/* Array features: out[i, :] = sum_{a < bag} table[ids[i * bag + a], :], added in the order of a
 * (`self.sparse_emb[k](tensor_feature).sum(2)`, embedding.py:247, :527, :584).  ids is (n, bag) int64; an id
 * outside [0, n_rows) adds nothing.  The backward pass is hmx_scatter_add_rows_f32 over the n * bag ids with each
 * output-gradient row repeated bag times. */
int hmx_gather_sum_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n, int64_t bag,
                            float* out, void* stream);

/* Pointwise activations between the projections.  mode 0: ReLU (F.relu embedding.py:606, nn.ReLU :118, :134);
 * mode 1: exact GELU x * Phi(x) (F.gelu embedding.py:644, nn.GELU :106, :110, :127).  Backward: dx = g * f'(x). */
int hmx_act_fwd(const float* x, int64_t n, int mode, float* out, void* stream);
int hmx_act_bwd(const float* x, const float* g, int64_t n, int mode, float* dx, void* stream);

/* Gated fusion (embedding.py:254-255): out = p * sigmoid(q), p = a feature's projection, q = its gate's linear
 * output.  Backward: dp = g * sigmoid(q) (the gate's own dependence on p arrives through q's producer),
 * dq = g * p * sigmoid(q) * (1 - sigmoid(q)). */
int hmx_gate_fwd(const float* p, const float* q, int64_t n, float* out, void* stream);
int hmx_gate_bwd(const float* p, const float* q, const float* g, int64_t n, float* dp, float* dq, void* stream);

/* Gated mix of the multi-scale fusion tail (hydra3.py:536-541: gate = sigmoid(fusion_gate linear); fused = gate *
 * attended + (1 - gate) * long_term): out = sigmoid(z) * a + (1 - sigmoid(z)) * l in one pass, z = the gate's linear
 * output.  Backward: dz = g (a - l) s (1 - s), da = g s, dl = g (1 - s) with s = sigmoid(z). */
int hmx_gate_mix_fwd(const float* z, const float* a, const float* l, int64_t n, float* out, void* stream);
int hmx_gate_mix_bwd(const float* z, const float* a, const float* l, const float* g, int64_t n, float* dz, float* da,
                     float* dl, void* stream);

/* nn.LayerNorm over the last dimension (embedding.py:56, :105, :109, :117, :126, :133): biased variance, eps inside
 * the root, y = (x - mean) * rstd * weight + bias; d <= 512.  The forward keeps mean and rstd (T each) for the
 * backward, which writes dx and, per workgroup b < hmx_layernorm_bwd_blocks(T), one partial row
 * part[b] = [dweight (d) | dbias (d)]; the parameter gradients are the column sums of part (hmx_colsum_f32). */
int64_t hmx_layernorm_bwd_blocks(int64_t T);
int hmx_layernorm_fwd(const float* x, const float* weight, const float* bias, int64_t T, int64_t d, float eps, float* y,
                      float* mean, float* rstd, void* stream);
int hmx_layernorm_bwd(const float* x, const float* weight, const float* mean, const float* rstd, const float* dy,
                      int64_t T, int64_t d, float* dx, float* part, void* stream);

/* Dropout multipliers (nn.Dropout at hydra3.py:411,445 and embedding.py:107,119; the attention-weight dropout of
 * nn.MultiheadAttention at hydra3.py:484 and embedding.py:51): out[i] = 0 with probability p_drop, else
 * 1 / (1 - p_drop); the consumer multiplies.  Philox4x32-10: out[i] is a function of (seed, offset, i) alone, so a
 * draw can be repeated; callers advance `offset` (any value) or change `seed` between draws.  0 <= p_drop <= 1. */
int hmx_dropout_multiplier_f32(uint64_t seed, uint64_t offset, int64_t n, float p_drop, float* out, void* stream);

/* out[t] = 1 / max(|x_t|_2, eps) for the rows of x (T, d): the row scale of F.normalize(x, p=2, dim=-1)
 * (embedding.py:575, the raw multimodal item vectors before their modality encoder).  The caller hands it to
 * hmx_linear_f32 / the weight-gradient calls as in_rowmask, so the normalised copy of the wide rows is never
 * written. */
int hmx_row_inv_norm_f32(const float* x, int64_t T, int64_t d, float eps, float* out, void* stream);

/* Attention core of nn.MultiheadAttention(batch_first=True) as the cross-modal fusion uses it (embedding.py:48-53,
 * :76-78: query = one modality's sequence, key = value = another's; no masks).  q (B * Lq, d) and kv (B * Lk, 2 d) =
 * [k | v] are the ALREADY PROJECTED inputs (in_proj through hmx_linear_f32), heads are column blocks of d / H:
 *   out[b, i, h-block] = sum_l softmax_l(<q_i, k_l> / sqrt(d / H)) * drop[b, h, i, l] * v_l.
 * drop (B, H, Lq, Lk) is the dropout multiplier of the attention weights (NULL in eval).  p (B, H, Lq, Lk) receives
 * the softmax weights for the backward pass, which writes dq (B * Lq, d), dkv (B * Lk, 2 d) and uses ds, wv
 * (B, H, Lq, Lk each) as scratch.  Lk <= 4096. */
int hmx_attn_fwd(const float* q, const float* kv, const float* drop, int64_t B, int64_t Lq, int64_t Lk, int64_t d,
                 int64_t H, float* p, float* out, void* stream);
int hmx_attn_bwd(const float* q, const float* kv, const float* drop, const float* p, const float* dout, int64_t B,
                 int64_t Lq, int64_t Lk, int64_t d, int64_t H, float* dq, float* dkv, float* ds, float* wv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HMX_H */
