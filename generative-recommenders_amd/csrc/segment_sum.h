// Deterministic "sort, then sum runs" scatter-add shared by the embedding backward and the loss backward.
//
// out[key] += sum over the entries carrying that key of value(entry), where value(entry) is produced by a
// source functor (a materialised row, or coefficient x row).  The entries are sorted by key with a stable radix
// sort, so within a key they stay in entry order; see segment_sum_staged_kernel for the fixed association order.
#pragma once
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "hmx_common.h"
#include "radix_sort.h"

namespace hmx_seg {
namespace {   // internal linkage: every translation unit that includes this header gets its own kernels

constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 16;  // 256 CUs x 16 resident blocks worth of grid-stride workers

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero();
template <>
__device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float& a, const float& b) { a += b; }


// n / d for 32-bit n and a divisor fixed per launch, as a multiply-high and two shifts (Granlund-Montgomery round-up
// form: exact for every 32-bit n); the hardware has no integer divide, and the sort's first pass divides per entry.
struct FastDiv {
    uint32_t m, s1, s2, d;
    FastDiv() : m(0), s1(0), s2(0), d(1) {}
    explicit FastDiv(uint32_t div) : d(div ? div : 1) {
        int l = 0;
        while (((uint64_t)1 << l) < d) ++l;
        m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
        s1 = l > 0 ? 1 : 0;
        s2 = l > 0 ? l - 1 : 0;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
};

// value(entry) = rows[entry]: the plain scatter-add of materialised gradient rows
struct PlainRows {
    using Val = uint32_t;               // what travels with the key through the sort: the entry number
    static constexpr int kPiece = 512;  // the sorted array is cut every kPiece entries (part of this op's contract)
    static constexpr bool kBalanced = false;
    const float* rows;
    __device__ __forceinline__ Val make_val(int64_t e) const { return (uint32_t)e; }
    __device__ __forceinline__ int64_t row_index(Val v) const { return (int64_t)v; }
    template <class V>
    __device__ __forceinline__ V weigh(Val, const V& v) const { return v; }
};

__device__ __forceinline__ float4 vscale(float a, const float4& v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }
__device__ __forceinline__ float vscale(float a, const float& v) { return a * v; }

// value(entry) = coef[i, col0 + s] * rows[i] with entry = i * per_row + s: the transposed product of a sparse
// coefficient matrix with a dense row matrix, never materialised
struct WeightedRows {
    // The coefficient rides through the sort next to the source row number, so the run loop reads it from the
    // sorted stream instead of chasing a random 4-byte load per entry.
    using Val = uint2;                  // (source row i, coefficient bits)
    static constexpr int kPiece = 256;  // (128: twice the pieces to join; 512: too few blocks to balance the groups)
    static constexpr bool kBalanced = true;    // runs here are hundreds of entries long: one group per block of kPiece
    const float* rows;      // (n_src, d)
    const float* coef;      // (n_src, coef_stride)
    int per_row, coef_stride, col0;
    FastDiv by_per_row;
    WeightedRows(const float* rows_, const float* coef_, int per_row_, int coef_stride_, int col0_)
        : rows(rows_), coef(coef_), per_row(per_row_), coef_stride(coef_stride_), col0(col0_), by_per_row((uint32_t)per_row_) {}
    __device__ __forceinline__ Val make_val(int64_t e) const {
        const uint32_t i = by_per_row.div((uint32_t)e), sidx = (uint32_t)e - i * (uint32_t)per_row;   // e < 2^32 (checked by the sort)
        return make_uint2(i, __float_as_uint(coef[(int64_t)i * coef_stride + col0 + sidx]));
    }
    __device__ __forceinline__ int64_t row_index(Val v) const { return (int64_t)v.x; }
    template <class V>
    __device__ __forceinline__ V weigh(Val sv, const V& v) const { return vscale(__uint_as_float(sv.y), v); }
};

inline int lpr_log2_for(int64_t cpr) {
    int l = 0;
    while ((1 << l) < cpr && l < 6) ++l;
    return l;
}

// Where the destination row of entry e comes from: an int64 array (torch.long ids), or an int32 array; with
// skip_self the int32 values are draws r in [0, rows - 1) of entry e's own row i = e / per_row, standing for the row
// r + (r >= i), i.e. a uniform draw over the rows other than i (hydra3.py:817-830) that was never written out.
struct IdSource {
    const int64_t* i64;
    const int32_t* i32;
    uint32_t per_row;
    int skip_self;
    FastDiv by_per_row;
    IdSource(const int64_t* ids) : i64(ids), i32(nullptr), per_row(1), skip_self(0) {}
    IdSource(const int32_t* ids, int64_t per_row_, int skip)
        : i64(nullptr), i32(ids), per_row((uint32_t)per_row_), skip_self(skip), by_per_row((uint32_t)per_row_) {}
    __device__ __forceinline__ int64_t at(int64_t e) const {
        if (i64 != nullptr) return i64[e];
        int64_t r = i32[e];
        if (skip_self) r += (r >= (int64_t)by_per_row.div((uint32_t)e)) ? 1 : 0;     // e < 2^32 (checked by the sort)
        return r;
    }
};

// ---------------------------------------------------------------- scatter-add ----------
// Steps 1 and 2: sort key = row id (or `sentinel` for entries that must not contribute), value = Src::make_val(i);
// the first pass of the sort reads them through this functor.
template <class Src>
struct EntriesFromIds {
    Src src;
    IdSource ids;
    const int64_t* token_type;
    int64_t want_type, n_rows, padding_idx;
    uint32_t sentinel;
    __device__ __forceinline__ uint32_t key(int64_t i) const {
        const int64_t id = ids.at(i);
        bool ok = id >= 0 && id < n_rows && id != padding_idx;
        if (token_type != nullptr) ok = ok && token_type[i] == want_type;
        return ok ? (uint32_t)id : sentinel;
    }
    __device__ __forceinline__ typename Src::Val val(int64_t i) const { return src.make_val(i); }
};

// Step 3 (pass 1): the sorted array is cut at every change of id AND at every multiple of kSuper entries into
// "pieces"; a piece is summed sequentially in sorted (= original, the sort is stable) order, starting from zero.
//   piece == whole run            -> added to the table row (exactly one writer per row)
//   piece continues an older run  -> head_piece[block]   (it starts at the block's first entry)
//   piece starts a run that goes on past the block end -> tail_piece[block]
// Pass 2: the group of the block where such a run starts adds tail_piece[b] + head_piece[b+1] + ... in
// block order and writes the row.  The association order depends only on positions in the sorted array,
// never on timing, so the result is bitwise reproducible; no float atomics; the sequential depth is
// bounded by kSuper whatever the id skew.
//
// One workgroup owns one block of kSuper sorted entries.  Its keys and values are read once into LDS; the
// wave ballots of "differs from the entry before" give every piece its number and extent (one entry in the
// list per distinct row of the block).  The source rows -- a random 4 * d bytes each, the bulk of the traffic --
// come in through the LDS DMA path in sub-blocks of S entries (32 KB, no registers held) into ONE stage per
// workgroup; the table rows of the sub-block's whole runs are requested just before its DMA, so one memory round
// trip serves both.  Overlap comes from the four workgroups a CU holds, not from a second stage: walking a
// sub-block out of LDS takes a fraction of its round trip, so about four fifths of the CU's stage bytes are in
// flight at any time (two stages per workgroup keep half of them in flight: measured 8 % slower on uniform ids,
// 20 % on Zipf ids).  A piece that crosses a sub-block boundary hands its running sum to the next sub-block
// through a carry row, so its additions still happen one after the other in entry order.
constexpr int kSuper = 512;
constexpr int kStageBytes = 32 * 1024;

template <int VEC, class Src>
__global__ __launch_bounds__(kBlock) void segment_sum_staged_kernel(Src src, const uint32_t* __restrict__ skeys,
                                                                    const typename Src::Val* __restrict__ svals,
                                                                    int64_t n, int cpr, int lpr_log2, int S,
                                                                    int stage_bytes, uint32_t sentinel,
                                                                    float* __restrict__ grad_table,
                                                                    uint8_t* __restrict__ row_flags,
                                                                    float* __restrict__ head_piece,
                                                                    float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    using Val = typename Src::Val;
    constexpr int CH = VEC * 4;               // bytes per lane per DMA instruction
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // the stage, then two carry rows
    __shared__ uint32_t keyL[kSuper + 1];     // [nb] holds the key after the block
    __shared__ Val valL[kSuper];
    __shared__ uint16_t piece_start[kSuper + 2];   // 16-bit lists: two workgroups of the 8-byte-value form fit a CU
    __shared__ uint16_t sub_first[kSuper + 2];     // first piece of each sub-block
    __shared__ int wave_cnt[2][kBlock / 64], wave_valid[2][kBlock / 64];
    __shared__ int carry_flag[2];             // "the carried piece began at a true run start"
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t sb = blockIdx.x;
    const int64_t j0 = sb * kSuper;
    if (skeys[j0] == sentinel) return;        // sentinels sort last: nothing in this block contributes
    const int nb = (int)((n - j0 < kSuper) ? n - j0 : kSuper);
    for (int i = tid; i < nb; i += kBlock) {
        keyL[i] = skeys[j0 + i];
        valL[i] = svals[j0 + i];
    }
    if (tid == 0) keyL[nb] = (j0 + nb < n) ? skeys[j0 + nb] : sentinel;
    const uint32_t key_before = (j0 > 0) ? skeys[j0 - 1] : sentinel;
    __syncthreads();

    // piece list: an entry starts a piece when its key differs from the entry before it, or when it is the first
    // of a sub-block (so that no piece spans two sub-blocks; the carry row joins the halves again)
    bool my_start[2];
    int my_rank[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = r * kBlock + tid;
        const bool valid = i < nb && keyL[i] != sentinel;
        const bool start = valid && (i % S == 0 || keyL[i] != keyL[i - 1]);
        const unsigned long long bs = __ballot(start), bv = __ballot(valid);
        my_start[r] = start;
        my_rank[r] = __popcll(bs & ((1ull << lane) - 1ull));
        if (lane == 0) {
            wave_cnt[r][w] = __popcll(bs);
            wave_valid[r][w] = __popcll(bv);
        }
    }
    __syncthreads();
    int n_pieces = 0, n_valid = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int base = 0;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int w2 = 0; w2 < kBlock / 64; ++w2) {
                if (r2 < r || (r2 == r && w2 < w)) base += wave_cnt[r2][w2];
                if (r == 0) {
                    n_pieces += wave_cnt[r2][w2];
                    n_valid += wave_valid[r2][w2];
                }
            }
        if (my_start[r]) {
            const int i = r * kBlock + tid;
            piece_start[base + my_rank[r]] = (uint16_t)i;
            if (i % S == 0) sub_first[i / S] = (uint16_t)(base + my_rank[r]);
        }
    }
    const int n_sub = (n_valid + S - 1) / S;
    if (tid == 0) {
        piece_start[n_pieces] = (uint16_t)n_valid;
        sub_first[n_sub] = (uint16_t)n_pieces;
    }
    __syncthreads();

    // DMA of one sub-block: flat 16-byte (4-byte when rows are not 16-byte multiples) chunk f of the stage is chunk
    // f % cpr of entry f / cpr; a wave's instruction covers 64 consecutive chunks and lands them dense in LDS.
    const int q0 = tid / cpr, r0 = tid % cpr, step_q = kBlock / cpr, step_r = kBlock % cpr;
    auto issue = [&](int k) {
        char* st = smem;
        const int e0 = k * S;
        const int cnt = (n_valid - e0 < S) ? n_valid - e0 : S;
        const int total = cnt * cpr;
        int q = q0, r = r0;
        for (int f0 = 0; f0 < total; f0 += kBlock) {
            if (f0 + tid < total) {
                const char* g = reinterpret_cast<const char*>(src.rows) + (src.row_index(valL[e0 + q]) * cpr + r) * CH;
                auto* gp = (const __attribute__((address_space(1))) void*)g;
                auto* lp = (__attribute__((address_space(3))) void*)(st + (f0 + w * 64) * CH);
                if constexpr (VEC == 4) __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                else __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
            }
            r += step_r;
            q += step_q;
            if (r >= cpr) {
                r -= cpr;
                ++q;
            }
        }
    };

    const int lpr = 1 << lpr_log2;
    const int lane_c = tid & (lpr - 1), grp = tid >> lpr_log2, n_groups = kBlock >> lpr_log2;
    const int cpr_up = (cpr + lpr - 1) & ~(lpr - 1);
    V* carry = reinterpret_cast<V*>(smem + stage_bytes);   // [2][cpr]
    for (int k = 0; k < n_sub; ++k) {
        __syncthreads();                                    // sub-block k - 1 has been consumed: the stage may be overwritten
        const V* st = reinterpret_cast<const V*>(smem);
        const int e0 = k * S;
        // U pieces per trip: their table rows are requested first (and before the next stage's DMA, which would
        // otherwise queue ahead of them), then each piece is summed out of LDS while those reads are in flight.
        constexpr int U = 4;
        const int p_end = sub_first[k + 1];
        bool dma_issued = false;
        for (int p0 = sub_first[k] + grp; p0 < p_end || !dma_issued; p0 += n_groups * U) {
            int s_pos[U], e_pos[U];
            uint32_t key[U];
            bool live[U], continues[U], goes_on[U], starts_true[U], whole[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * n_groups;
                live[u] = p < p_end;
                s_pos[u] = live[u] ? piece_start[p] : 0;
                e_pos[u] = live[u] ? piece_start[p + 1] : 0;
                key[u] = keyL[s_pos[u]];
                continues[u] = s_pos[u] == e0 && k > 0 && keyL[s_pos[u] > 0 ? s_pos[u] - 1 : 0] == key[u];   // other half: carry row
                goes_on[u] = e_pos[u] == e0 + S && e_pos[u] < n_valid && keyL[e_pos[u]] == key[u];
                starts_true[u] = continues[u] ? (carry_flag[(k - 1) & 1] != 0) : (s_pos[u] > 0 || key_before != key[u]);
                whole[u] = live[u] && !goes_on[u] && starts_true[u] && keyL[e_pos[u]] != key[u];
            }
            // every lane walks the padded row width, so the first trip's first chunk is one point that each wave passes
            // exactly once per stage, whatever the row width: the DMA is issued there (and, with one buffer, awaited)
            for (int c = lane_c; c < cpr_up; c += lpr) {
                const bool ok = c < cpr;
                V cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    cur[u] = vzero<VEC>();
                    if (whole[u] && ok) cur[u] = (reinterpret_cast<const V*>(grad_table) + (int64_t)key[u] * cpr)[c];
                }
                asm volatile("" ::: "memory");
                if (!dma_issued) {
                    issue(k);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the sub-block is in LDS
                    __syncthreads();                                    // everyone's is
                    dma_issued = true;
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!live[u] || !ok) continue;
                    V acc = continues[u] ? carry[((k - 1) & 1) * cpr + c] : vzero<VEC>();
                    const V* rowp = st + (s_pos[u] - e0) * cpr + c;
                    int e = s_pos[u];
                    // long pieces: sixteen LDS row reads in flight, so the chain of additions, not the LDS latency,
                    // sets the pace of a run that fills the stage
                    for (; e + 16 <= e_pos[u]; e += 16, rowp += 16 * cpr) {
                        V v[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = src.weigh(valL[e + i], rowp[i * cpr]);
#pragma unroll
                        for (int i = 0; i < 16; ++i) vadd(acc, v[i]);
                    }
                    for (; e + 4 <= e_pos[u]; e += 4, rowp += 4 * cpr) {
                        const V v0 = rowp[0], v1 = rowp[cpr], v2 = rowp[2 * cpr], v3 = rowp[3 * cpr];
                        vadd(acc, src.weigh(valL[e], v0));
                        vadd(acc, src.weigh(valL[e + 1], v1));
                        vadd(acc, src.weigh(valL[e + 2], v2));
                        vadd(acc, src.weigh(valL[e + 3], v3));
                    }
                    for (; e < e_pos[u]; ++e, rowp += cpr) vadd(acc, src.weigh(valL[e], rowp[0]));
                    if (goes_on[u]) {
                        carry[(k & 1) * cpr + c] = acc;
                    } else if (whole[u]) {
                        vadd(cur[u], acc);
                        (reinterpret_cast<V*>(grad_table) + (int64_t)key[u] * cpr)[c] = cur[u];
                        if (row_flags != nullptr && c == 0) row_flags[key[u]] = 1;
                    } else if (!starts_true[u]) {
                        reinterpret_cast<V*>(head_piece)[sb * cpr + c] = acc;
                    } else {
                        reinterpret_cast<V*>(tail_piece)[sb * cpr + c] = acc;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (live[u] && goes_on[u] && lane_c == 0) carry_flag[k & 1] = starts_true[u] ? 1 : 0;
        }
    }
}

// Pass 1, balanced form: one group walks one block of SB consecutive sorted entries, whatever runs it holds, so
// every group has the same amount of work and long runs are summed by many groups at once.  Partial sums of runs
// that cross a block boundary go to head_piece / tail_piece exactly as in segment_sum_staged_kernel (with kSuper = SB);
// runs that lie inside one block are added to their row directly (one writer per row).
template <int VEC, class Src, int SB>
__global__ __launch_bounds__(kBlock) void segment_sum_balanced_kernel(Src src, const uint32_t* __restrict__ skeys,
                                                                      const typename Src::Val* __restrict__ svals,
                                                                      int64_t n, int cpr, int lpr_log2,
                                                                      uint32_t sentinel, float* __restrict__ grad_table,
                                                                      uint8_t* __restrict__ row_flags,
                                                                      float* __restrict__ head_piece,
                                                                      float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    constexpr int U = 8;   // entries requested per trip (summed in order)
    static_assert(SB % U == 0, "block size must be a multiple of the prefetch depth");
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int groups_per_block = kBlock >> lpr_log2;
    const int64_t n_sb = (n + SB - 1) / SB;
    for (int64_t b = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> lpr_log2); b < n_sb;
         b += (int64_t)gridDim.x * groups_per_block) {
        const int64_t j0 = b * SB;
        const int64_t j1 = (j0 + SB < n) ? j0 + SB : n;
        const uint32_t key_before = (j0 > 0) ? skeys[j0 - 1] : sentinel;
        const uint32_t key_after = (j1 < n) ? skeys[j1] : sentinel;
        for (int c = lane_c; c < cpr; c += lpr) {
            V acc = vzero<VEC>();
            uint32_t cur = skeys[j0];
            bool first_run = true;   // the run being summed began at j0 (it may continue an older block's run)
            auto flush = [&](bool ends_here) {
                if (cur == sentinel) return;
                const bool starts_true = !(first_run && cur == key_before);
                V* dst;
                if (starts_true && ends_here) {
                    dst = reinterpret_cast<V*>(grad_table) + (int64_t)cur * cpr + c;
                    V t = *dst;
                    vadd(t, acc);
                    *dst = t;
                    if (row_flags != nullptr && c == 0) row_flags[cur] = 1;
                } else if (!starts_true) {
                    reinterpret_cast<V*>(head_piece)[b * cpr + c] = acc;
                } else {
                    reinterpret_cast<V*>(tail_piece)[b * cpr + c] = acc;
                }
            };
            // Keys and values of a trip are requested one trip ahead, AFTER the row loads of the trip before: loads
            // return in issue order, so waiting for those rows leaves the next keys and values in flight, and a trip
            // costs one memory round trip (its rows) instead of two (keys / values, then rows).
            uint32_t key[U], nkey[U];
            typename Src::Val sv[U], nsv[U];
            auto fetch_entries = [&](uint32_t(&k)[U], typename Src::Val(&w)[U], int64_t e) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = e + u < j1;
                    k[u] = in ? skeys[e + u] : sentinel;
                    if (in) w[u] = svals[e + u];
                }
            };
            fetch_entries(key, sv, j0);
            for (int64_t e = j0; e < j1; e += U) {
                V v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = (key[u] != sentinel)
                               ? src.weigh(sv[u], reinterpret_cast<const V*>(src.rows)[src.row_index(sv[u]) * cpr + c])
                               : vzero<VEC>();
                if (e + U < j1) fetch_entries(nkey, nsv, e + U);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (e + u >= j1) break;
                    if (key[u] != cur) {
                        flush(true);
                        acc = vzero<VEC>();
                        cur = key[u];
                        first_run = false;
                    }
                    if (key[u] != sentinel) vadd(acc, v[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    key[u] = nkey[u];
                    sv[u] = nsv[u];
                }
            }
            flush(cur != key_after);
        }
    }
}

// Pass 2: one group per block b of SB entries; acts only if a run starts inside b and continues past its end.
// The pieces of a run are added strictly in order (that is the summation contract), so a run of many pieces is a
// serial chain; what can be taken off that chain is the memory latency.  A run that covers more than kJoinLong further
// blocks (skewed ids: one popular row may hold a tenth of all entries) is therefore joined by its whole wave: the
// run's last entry is found by a 64-way search over the sorted keys (no per-piece key probes), all lanes fetch the
// next kJoinFetch * (groups per wave) head pieces together into LDS, and the first group adds them in order from
// there (8.4 M Zipf(1.05) ids, 4 M x 64: 696 -> 60 us for this kernel).
constexpr int kJoinLong = 8;       // further blocks from which the wave-cooperative form is used
constexpr int kJoinFetch = 8;      // head pieces per lane group and round
constexpr int kJoinLdsPerWave = 8 * 1024;

template <int VEC, int SB>
__global__ __launch_bounds__(kBlock) void segment_join_kernel(const uint32_t* __restrict__ skeys, int64_t n, int cpr,
                                                              int lpr_log2, uint32_t sentinel,
                                                              float* __restrict__ grad_table,
                                                              uint8_t* __restrict__ row_flags,
                                                              const float* __restrict__ head_piece,
                                                              const float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    extern __shared__ __attribute__((aligned(16))) char join_smem[];
    constexpr int U = 8;   // head pieces fetched per trip (their "does the run go on" keys with them)
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int groups_per_block = kBlock >> lpr_log2;
    const int64_t n_sb = (n + SB - 1) / SB;
    const bool coop_ok = VEC == 4 && cpr <= lpr;       // a row is one V per lane of a group
    const int lane = threadIdx.x & 63;
    const int G = 64 >> lpr_log2;                      // groups per wave
    const int64_t stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t first = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> lpr_log2);
    // every lane of a wave makes the same number of trips (the cooperative form needs the whole wave)
    const int64_t wave_first = (int64_t)blockIdx.x * groups_per_block + ((threadIdx.x >> 6) << 6 >> lpr_log2);
    for (int64_t b0 = wave_first; b0 < n_sb; b0 += stride) {
        const int64_t b = b0 + (first - wave_first);
        bool act = false;
        uint32_t key = sentinel;
        if (b < n_sb) {
            const int64_t end = (b + 1) * SB;
            if (end < n) {                                            // last block: nothing continues
                key = skeys[end - 1];
                act = key != sentinel && skeys[end] == key &&          // a run crosses this boundary ...
                      !(skeys[b * SB] == key && b > 0 && skeys[b * SB - 1] == key);   // ... and it started in this block
            }
        }
        bool is_long = false;
        if (act && coop_ok) {
            const int64_t far = (b + 1 + kJoinLong) * SB;
            is_long = far < n && skeys[far] == key;                   // sorted keys: everything in between is the run
        }
        if (act && !is_long) {
            for (int c = lane_c; c < cpr; c += lpr) {
                V acc = reinterpret_cast<const V*>(tail_piece)[b * cpr + c];
                int64_t cb = b + 1;       // block cb is known to begin with this run
                bool more = true;
                while (more) {
                    V hp[U];
                    bool goes_on[U];      // the run also covers the start of block cb + u + 1
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int64_t cend = (cb + u + 1) * SB;
                        goes_on[u] = cend < n && skeys[cend - 1] == key && skeys[cend] == key;
                        hp[u] = (cb + u < n_sb) ? reinterpret_cast<const V*>(head_piece)[(cb + u) * cpr + c] : vzero<VEC>();
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!more) break;
                        vadd(acc, hp[u]);
                        more = goes_on[u];
                    }
                    cb += U;
                }
                V* dst = reinterpret_cast<V*>(grad_table) + (int64_t)key * cpr + c;
                V cur = *dst;
                vadd(cur, acc);
                *dst = cur;
                if (row_flags != nullptr && c == 0) row_flags[key] = 1;
            }
        }
        if constexpr (VEC == 4) {
            unsigned long long todo = __ballot(is_long && lane_c == 0);
            V* stage = reinterpret_cast<V*>(join_smem + (size_t)(threadIdx.x >> 6) * kJoinLdsPerWave);
            const int g = lane >> lpr_log2;
            const int P = G * kJoinFetch;                               // pieces per round
            while (todo) {
                const int src_lane = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int64_t bb = b0 + (src_lane >> lpr_log2);
                const uint32_t rkey = (uint32_t)__builtin_amdgcn_readlane((int)key, src_lane);
                // last entry of the run: 64-way search in the sorted keys; skeys[lo] == rkey throughout
                int64_t lo = (bb + 1) * SB, hi = n;
                while (hi - lo > 1) {
                    const int64_t step = (hi - lo + 63) / 64;
                    const int64_t idx = lo + (int64_t)(lane + 1) * step;
                    const bool hit = idx < hi && skeys[idx] == rkey;
                    const int t = __popcll(__ballot(hit));              // sorted: the hits are lanes 0 .. t - 1
                    hi = (lo + (int64_t)(t + 1) * step < hi) ? lo + (int64_t)(t + 1) * step : hi;
                    lo += (int64_t)t * step;
                }
                const int64_t cbe = lo / SB;                            // last block that begins with this run
                V acc = vzero<VEC>();
                if (g == 0 && lane_c < cpr) acc = reinterpret_cast<const V*>(tail_piece)[bb * cpr + lane_c];
                for (int64_t base = bb + 1; base <= cbe; base += P) {
                    const int cnt = (int)((cbe - base + 1 < P) ? cbe - base + 1 : P);
                    V regs[kJoinFetch];
#pragma unroll
                    for (int i = 0; i < kJoinFetch; ++i) {
                        const int pidx = g + G * i;
                        regs[i] = (pidx < cnt && lane_c < cpr)
                                      ? reinterpret_cast<const V*>(head_piece)[(base + pidx) * cpr + lane_c] : vzero<VEC>();
                    }
#pragma unroll
                    for (int i = 0; i < kJoinFetch; ++i) {
                        const int pidx = g + G * i;
                        if (pidx < cnt && lane_c < cpr) stage[pidx * cpr + lane_c] = regs[i];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (g == 0 && lane_c < cpr)
                        for (int i = 0; i < cnt; ++i) vadd(acc, stage[i * cpr + lane_c]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (g == 0 && lane_c < cpr) {
                    V* dst = reinterpret_cast<V*>(grad_table) + (int64_t)rkey * cpr + lane_c;
                    V cur = *dst;
                    vadd(cur, acc);
                    *dst = cur;
                    if (row_flags != nullptr && lane_c == 0) row_flags[rkey] = 1;
                }
            }
        }
    }
}

inline int sort_bits(int64_t n_rows) {
    int b = 1;
    while (((int64_t)1 << b) <= n_rows) ++b;  // sentinel == n_rows must be representable
    return b;
}

constexpr int kMaxRowFloats = kStageBytes / 4;      // one row must fit an LDS stage
constexpr int kMaxStagedLds = kStageBytes + 2 * kStageBytes;   // one stage + two carry rows at the widest row

inline size_t piece_bytes(int64_t n, int64_t d, int piece) {
    return hmx_align_up((size_t)hmx_ceil_div(n, piece) * (size_t)d * sizeof(float), 256);
}

template <class Src>
inline size_t workspace_bytes(int64_t n, int64_t n_rows, int64_t d) {
    if (n <= 0 || n_rows <= 0 || d <= 0) return 256;
    const size_t karr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    const size_t varr = hmx_align_up((size_t)n * sizeof(typename Src::Val), 256);
    return 2 * karr + 2 * varr + 2 * piece_bytes(n, d, Src::kPiece) + hmx_sort::counter_bytes(n) + 256;
}

// table[ids[e]] += value(e) for e in [0, n).  `vec`: rows and table are 16-byte aligned and d % 4 == 0.
// sorted_keys_out (optional) receives the device pointer of the sorted keys inside the workspace (n entries,
// `sentinel_out` marks entries that did not contribute; they sort last).
// row_flags (optional, n_rows bytes): row_flags[r] = 1 is stored for every table row that receives a sum, so that a
// consumer of a mostly-zero dense gradient (the optimizer, the data-parallel exchange) knows which rows to read.
template <class Src>
int sorted_segment_add(const Src& src, bool vec, const IdSource& ids, const int64_t* token_type, int64_t want_type,
                       int64_t n, int64_t d, float* table, int64_t n_rows, int64_t padding_idx, void* workspace,
                       size_t ws_bytes, hipStream_t s, const uint32_t** sorted_keys_out = nullptr,
                       uint32_t* sentinel_out = nullptr, uint8_t* row_flags = nullptr) {
    if (d > kMaxRowFloats) return HMX_EINVAL;
    const int bits = sort_bits(n_rows);
    using Val = typename Src::Val;
    if (ws_bytes < workspace_bytes<Src>(n, n_rows, d) - 256) return HMX_EWORKSPACE;
    HmxCarver ws(workspace);
    uint32_t* keys = ws.take<uint32_t>(n);
    Val* vals = ws.take<Val>(n);
    uint32_t* skeys = ws.take<uint32_t>(n);
    Val* svals = ws.take<Val>(n);
    float* head_piece = ws.take<float>(piece_bytes(n, d, Src::kPiece) / sizeof(float));
    float* tail_piece = ws.take<float>(piece_bytes(n, d, Src::kPiece) / sizeof(float));
    uint32_t* counters = ws.take<uint32_t>(hmx_sort::counter_bytes(n) / sizeof(uint32_t));
    const uint32_t sentinel = (uint32_t)n_rows;
    if (sorted_keys_out) *sorted_keys_out = skeys;
    if (sentinel_out) *sentinel_out = sentinel;

    const EntriesFromIds<Src> first{src, ids, token_type, want_type, n_rows, padding_idx, sentinel};
    const int rc = hmx_sort::sort_entries(first, n, bits, skeys, svals, keys, vals, counters, s);
    if (rc != HMX_OK) return rc;
    HMX_CHECK_LAUNCH();

    const int64_t cpr = vec ? d / 4 : d;
    const int l2 = lpr_log2_for(cpr);
    static_assert(Src::kBalanced || Src::kPiece == kSuper, "the staged kernel cuts at kSuper");
    constexpr int SB = Src::kPiece;
    const int64_t n_sb = hmx_ceil_div(n, SB);
    const int row_bytes = (int)d * 4;
    const int S = std::max(1, std::min(kBlock, std::max(kStageBytes, row_bytes) / row_bytes));   // entries per LDS stage
    const int stage_bytes = (int)hmx_align_up((size_t)S * row_bytes, 1024);
    const int lds = stage_bytes + 2 * (int)hmx_align_up((size_t)row_bytes, 16);
    const int grid3 = (int)std::min<int64_t>(hmx_ceil_div(n_sb, kBlock >> l2), (int64_t)kMaxGrid * 4);
    const int grid4 = (int)std::min<int64_t>(hmx_ceil_div(n_sb, kBlock >> l2), (int64_t)kMaxGrid);
    auto run = [&](auto vec_tag) {
        constexpr int VEC = decltype(vec_tag)::value;
        if constexpr (Src::kBalanced) {
            hipLaunchKernelGGL((segment_sum_balanced_kernel<VEC, Src, SB>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals,
                               n, (int)cpr, l2, sentinel, table, row_flags, head_piece, tail_piece);
        } else {
            hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&segment_sum_staged_kernel<VEC, Src>), kMaxStagedLds);
            hipLaunchKernelGGL((segment_sum_staged_kernel<VEC, Src>), dim3((unsigned)n_sb), dim3(kBlock), lds, s, src, skeys,
                               svals, n, (int)cpr, l2, S, stage_bytes, sentinel, table, row_flags, head_piece, tail_piece);
        }
        hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&segment_join_kernel<VEC, SB>), (kBlock / 64) * kJoinLdsPerWave);
        hipLaunchKernelGGL((segment_join_kernel<VEC, SB>), dim3(grid4), dim3(kBlock), (kBlock / 64) * kJoinLdsPerWave, s, skeys,
                           n, (int)cpr, l2, sentinel, table, row_flags, head_piece, tail_piece);
    };
    if (vec) run(std::integral_constant<int, 4>{});
    else run(std::integral_constant<int, 1>{});
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

}  // namespace
}  // namespace hmx_seg
