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

/* Token-type routed gather (embedding.py:436-450):
 *   out[i] = item_table[ids[i]] if token_type[i]==1, user_table[ids[i]] if ==2, else 0. */
int hmx_gather_routed_f32(const float* item_table, int64_t item_rows, const float* user_table, int64_t user_rows,
                          int64_t d, const int64_t* ids, const int64_t* token_type, int64_t n, float* out,
                          void* stream);

/* ---------------------------------------------------------------------------------
 * Scatter-add of gradient rows (backward of the gathers above).
 *   grad_table[ids[i], :] += grad_rows[i, :]   for every i with ids[i] != padding_idx,
 *   ids[i] in range, and (token_type == NULL or token_type[i] == want_type).
 * Replaces ATen embedding_dense_backward / index_put_(accumulate=True) reached from
 * main.py:708 through embedding.py:445,450,454 and hydra3.py:917.
 * Deterministic: contributions to one row are summed in increasing i (a stable sort by id
 * followed by one sequential segment sum per distinct id; no float atomics).
 * grad_table is NOT cleared here (the caller zero-fills or accumulates).
 * ------------------------------------------------------------------------------- */
size_t hmx_scatter_add_rows_workspace_bytes(int64_t n, int64_t n_rows);
int hmx_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                             int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                             int64_t padding_idx, void* workspace, size_t workspace_bytes, void* stream);

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

/* Merge S per-shard top-K lists (cand_* laid out (S, n_q, K), as RCCL all_gather leaves
 * them) into the global top-K with the same order / tie-break.  Replaces nothing in the
 * reference (it is single-device); it is the exchange step of item-range sharding. */
int hmx_topk_merge(const int64_t* cand_ids, const float* cand_scores, int64_t n_shards, int64_t n_q, int64_t k,
                   int64_t* out_ids, float* out_scores, void* stream);

/* ---------------------------------------------------------------------------------
 * Benchmark support (no counterpart in the reference).  When enabled, entry points bracket
 * their dominant kernel with HIP events on the caller's stream; hmx_profile_collect blocks
 * until the events of spans tagged `tag` have completed, returns their summed duration and
 * count, and forgets them.  Tags: "topk_coarse_bf16", "topk_coarse_f32".
 * ------------------------------------------------------------------------------- */
int hmx_profile_enable(int on);
int hmx_profile_collect(const char* tag, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* HMX_H */
