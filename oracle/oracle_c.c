/*
 * CPU oracle for the integer / index / exact-order parts of the hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, the smoke check and the
 * CPU-baseline leg of the benchmark load this library.
 *
 * Parity: pinned by tests/test_oracle_golden.py and tests/test_c_oracle.py against
 * golden vectors captured from the reference (tests/golden/g9_topk.npz: ids from the
 * reference's torch.mm + torch.topk, hydra3.py:993-1006) and against PyTorch's own
 * nn.Embedding forward / backward on CPU (embedding.py:157-164, 436-454).
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -march=native -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[i] = table[ids[i]]; an id outside [0, n_rows) reads as a zero row.  (embedding.py:454) */
void orc_gather_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t id = ids[i];
        if (id >= 0 && id < n_rows) memcpy(out + i * d, table + id * d, (size_t)d * sizeof(float));
        else memset(out + i * d, 0, (size_t)d * sizeof(float));
    }
}

/* embedding.py:436-450: token type 1 -> item table, 2 -> user table, anything else -> zeros. */
void orc_gather_routed_f32(const float* item_table, int64_t item_rows, const float* user_table, int64_t user_rows,
                           int64_t d, const int64_t* ids, const int64_t* token_type, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t id = ids[i];
        const float* src = NULL;
        if (token_type[i] == 1 && id >= 0 && id < item_rows) src = item_table + id * d;
        if (token_type[i] == 2 && id >= 0 && id < user_rows) src = user_table + id * d;
        if (src) memcpy(out + i * d, src, (size_t)d * sizeof(float));
        else memset(out + i * d, 0, (size_t)d * sizeof(float));
    }
}

/* Backward of the gathers (ATen embedding_dense_backward / index_put_ accumulate, reached from
 * main.py:708).  Summation order (the contract of the HIP kernel): contributing entries are ordered by
 * (id, i); that ordered list is cut at every change of id and at every multiple of 512 positions; each
 * piece is summed from zero in order, the pieces of one id are added in order, the total is added to the
 * row once.  For an id with few occurrences this is simply the sum in increasing i. */
#define ORC_SUPER 512
void orc_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                              int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                              int64_t padding_idx) {
    int64_t* start = (int64_t*)calloc((size_t)n_rows + 1, sizeof(int64_t));
    int64_t n_valid = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t id = ids[i];
        if (id < 0 || id >= n_rows || id == padding_idx) continue;
        if (token_type && token_type[i] != want_type) continue;
        start[id + 1]++;
        n_valid++;
    }
    for (int64_t r = 0; r < n_rows; ++r) start[r + 1] += start[r];
    int64_t* order = (int64_t*)malloc((size_t)(n_valid > 0 ? n_valid : 1) * sizeof(int64_t));
    int64_t* fill = (int64_t*)calloc((size_t)n_rows, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {                     /* stable counting sort by id */
        int64_t id = ids[i];
        if (id < 0 || id >= n_rows || id == padding_idx) continue;
        if (token_type && token_type[i] != want_type) continue;
        order[start[id] + fill[id]++] = i;
    }
    float* total = (float*)malloc((size_t)d * sizeof(float));
    float* piece = (float*)malloc((size_t)d * sizeof(float));
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t s = start[r], e = start[r + 1];
        if (s == e) continue;
        int first = 1;
        while (s < e) {
            int64_t cut = (s / ORC_SUPER + 1) * ORC_SUPER;
            int64_t pe = cut < e ? cut : e;
            for (int64_t c = 0; c < d; ++c) piece[c] = 0.0f;
            for (int64_t p = s; p < pe; ++p)
                for (int64_t c = 0; c < d; ++c) piece[c] += grad_rows[order[p] * d + c];
            if (first) memcpy(total, piece, (size_t)d * sizeof(float));
            else for (int64_t c = 0; c < d; ++c) total[c] += piece[c];
            first = 0;
            s = pe;
        }
        for (int64_t c = 0; c < d; ++c) grad_table[r * d + c] += total[c];
    }
    free(start); free(order); free(fill); free(total); free(piece);
}

/* s(q, j) = fma chain over k = 0..d-1 in fp32, starting from 0. */
static inline float dot_fma_chain(const float* q, const float* t, int64_t d) {
    float acc = 0.0f;
    for (int64_t k = 0; k < d; ++k) acc = fmaf(q[k], t[k], acc);
    return acc;
}

/* (score desc, id asc) strict "a ranks before b" */
static inline int ranks_before(float sa, int64_t ia, float sb, int64_t ib) { return sa > sb || (sa == sb && ia < ib); }

static void insert_topk(float* sc, int64_t* id, int64_t k, int64_t* count, float s, int64_t i) {
    int64_t pos = *count;
    if (pos == k) {
        if (!ranks_before(s, i, sc[k - 1], id[k - 1])) return;
        pos = k - 1;
    } else {
        (*count)++;
    }
    while (pos > 0 && ranks_before(s, i, sc[pos - 1], id[pos - 1])) {
        sc[pos] = sc[pos - 1];
        id[pos] = id[pos - 1];
        --pos;
    }
    sc[pos] = s;
    id[pos] = i;
}

/* Exact scoring + top-k (hydra3.py:993-1006 with an explicit tie-break; k = 10 is the retrieval of
 * infer.py:405-417).  exclude: n_q x n_excl global ids scored as -inf (may be NULL).  Table row r has
 * global id id_offset + r.  Unfilled slots: id -1, score -inf. */
void orc_score_topk_f32(const float* queries, int64_t n_q, const float* table, int64_t n_rows, int64_t d, int64_t k,
                        const int64_t* exclude, int64_t n_excl, int64_t id_offset, int64_t* out_ids,
                        float* out_scores) {
#pragma omp parallel
    {
        float* scores = (float*)malloc((size_t)n_rows * sizeof(float));
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < n_q; ++q) {
            const float* qv = queries + q * d;
            for (int64_t r = 0; r < n_rows; ++r) scores[r] = dot_fma_chain(qv, table + r * d, d);
            if (exclude)
                for (int64_t e = 0; e < n_excl; ++e) {
                    int64_t r = exclude[q * n_excl + e] - id_offset;
                    if (r >= 0 && r < n_rows) scores[r] = -INFINITY;
                }
            float* sc = out_scores + q * k;
            int64_t* id = out_ids + q * k;
            int64_t count = 0;
            for (int64_t r = 0; r < n_rows; ++r) {
                if (scores[r] == -INFINITY) continue;
                if (count == k && !(scores[r] > sc[k - 1])) continue; /* ids ascend: an equal score never displaces */
                insert_topk(sc, id, k, &count, scores[r], id_offset + r);
            }
            for (int64_t s = count; s < k; ++s) {
                sc[s] = -INFINITY;
                id[s] = -1;
            }
        }
        free(scores);
    }
}

/* Merge per-shard lists laid out (S, n_q, k) into (n_q, k) with the same order / tie-break. */
void orc_topk_merge(const int64_t* cand_ids, const float* cand_scores, int64_t n_shards, int64_t n_q, int64_t k,
                    int64_t* out_ids, float* out_scores) {
    for (int64_t q = 0; q < n_q; ++q) {
        float* sc = out_scores + q * k;
        int64_t* id = out_ids + q * k;
        int64_t count = 0;
        for (int64_t s = 0; s < n_shards; ++s)
            for (int64_t j = 0; j < k; ++j) {
                int64_t ci = cand_ids[(s * n_q + q) * k + j];
                float cs = cand_scores[(s * n_q + q) * k + j];
                if (ci < 0) continue;
                insert_topk(sc, id, k, &count, cs, ci);
            }
        for (int64_t s = count; s < k; ++s) {
            sc[s] = -INFINITY;
            id[s] = -1;
        }
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
