// Deterministic "sort, then sum runs" scatter-add shared by the embedding backward and the loss backward.
//
// out[key] += sum over the entries carrying that key of value(entry), where value(entry) is produced by a
// source functor (a materialised row, or coefficient x row).  The entries are sorted by key with a stable radix
// sort, so within a key they stay in entry order; see segment_sum_kernel for the fixed association order.
#pragma once
#include <algorithm>
#include <cstring>
#include <type_traits>

#include <rocprim/device/device_radix_sort.hpp>

#include "hmx_common.h"

#ifndef HMX_PLAIN_IN_FLIGHT
#define HMX_PLAIN_IN_FLIGHT 16
#endif

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


// value(entry) = rows[entry]: the plain scatter-add of materialised gradient rows
struct PlainRows {
    using Val = uint32_t;               // what travels with the key through the sort: the entry number
    static constexpr int kInFlight = HMX_PLAIN_IN_FLIGHT; // entries requested per trip of the run loop (summed in order)
    static constexpr int kPiece = 512;  // the sorted array is also cut every kPiece entries
    static constexpr bool kBalanced = false;   // one group per run start (summation order fixed by the contract)
    const float* rows;
    __device__ __forceinline__ Val make_val(int64_t e) const { return (uint32_t)e; }
    template <int VEC>
    __device__ __forceinline__ typename VecT<VEC>::type load(Val pos, int cpr, int c) const {
        return reinterpret_cast<const typename VecT<VEC>::type*>(rows)[(int64_t)pos * cpr + c];
    }
};

__device__ __forceinline__ float4 vscale(float a, const float4& v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }
__device__ __forceinline__ float vscale(float a, const float& v) { return a * v; }

// value(entry) = coef[i, col0 + s] * rows[i] with entry = i * per_row + s: the transposed product of a sparse
// coefficient matrix with a dense row matrix, never materialised
struct WeightedRows {
    // The coefficient rides through the sort next to the source row number, so the run loop reads it from the
    // sorted stream instead of chasing a random 4-byte load per entry.
    using Val = uint2;                  // (source row i, coefficient bits)
    static constexpr int kInFlight = 8;
    static constexpr int kPiece = 128;
    static constexpr bool kBalanced = true;    // one group per block of kPiece entries (runs here are long)
    const float* rows;      // (n_src, d)
    const float* coef;      // (n_src, coef_stride)
    int per_row, coef_stride, col0;
    __device__ __forceinline__ Val make_val(int64_t e) const {
        const int64_t i = e / per_row, sidx = e - i * per_row;
        return make_uint2((uint32_t)i, __float_as_uint(coef[i * coef_stride + col0 + sidx]));
    }
    template <int VEC>
    __device__ __forceinline__ typename VecT<VEC>::type load(Val v, int cpr, int c) const {
        return vscale(__uint_as_float(v.y),
                      reinterpret_cast<const typename VecT<VEC>::type*>(rows)[(int64_t)v.x * cpr + c]);
    }
};

inline int lpr_log2_for(int64_t cpr) {
    int l = 0;
    while ((1 << l) < cpr && l < 6) ++l;
    return l;
}

// ---------------------------------------------------------------- scatter-add ----------
// Step 1: sort key = row id (or `sentinel` for entries that must not contribute), value = i.
template <class Src>
__global__ __launch_bounds__(kBlock) void make_keys_kernel(Src src, const int64_t* __restrict__ ids,
                                                           const int64_t* __restrict__ token_type, int64_t want_type,
                                                           int64_t n, int64_t n_rows, int64_t padding_idx,
                                                           uint32_t sentinel, uint32_t* __restrict__ keys,
                                                           typename Src::Val* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t id = ids[i];
        bool ok = id >= 0 && id < n_rows && id != padding_idx;
        if (token_type != nullptr) ok = ok && token_type[i] == want_type;
        keys[i] = ok ? (uint32_t)id : sentinel;
        vals[i] = src.make_val(i);
    }
}

// Step 3 (pass 1): the sorted array is cut at every change of id AND at every multiple of kSuper
// entries into "pieces"; one group of LPR lanes sums one piece sequentially in sorted (= original, the
// sort is stable) order.  A group owns the pieces whose first entry lies in its chunk of kChunk entries.
//   piece == whole run            -> added to the table row (exactly one writer per row)
//   piece continues an older run  -> head_piece[super-block]   (it starts at the block's first entry)
//   piece starts a run that goes on past the block end -> tail_piece[super-block]
// Pass 2: the group of the block where such a run starts adds tail_piece[b] + head_piece[b+1] + ... in
// block order and writes the row.  The association order depends only on positions in the sorted array,
// never on timing, so the result is bitwise reproducible; no float atomics; the sequential depth is
// bounded by kSuper whatever the id skew.
constexpr int kChunk = 8;
constexpr int kSuper = 512;

// WIDE: trips of Src::kInFlight entries with all keys requested up front, for inputs whose runs are long (a small
// table hit by many ids: the trip is three dependent memory latencies deep, so depth is what counts); otherwise
// trips of four with the key loads stopping at the run's end, which costs the least registers and loads when
// most runs are one or two entries long.  The additions happen in entry order either way: same bits.
template <int VEC, class Src, bool WIDE>
__global__ __launch_bounds__(kBlock) void segment_sum_kernel(Src src,
                                                             const uint32_t* __restrict__ skeys,
                                                             const typename Src::Val* __restrict__ svals, int64_t n, int cpr,
                                                             int lpr_log2, uint32_t sentinel,
                                                             float* __restrict__ grad_table,
                                                             float* __restrict__ head_piece,
                                                             float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int groups_per_block = kBlock >> lpr_log2;
    const int64_t n_groups = (n + kChunk - 1) / kChunk;
    for (int64_t g = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> lpr_log2); g < n_groups;
         g += (int64_t)gridDim.x * groups_per_block) {
        const int64_t j0 = g * kChunk;
        const int64_t j1 = (j0 + kChunk < n) ? j0 + kChunk : n;
        const int64_t sb = j0 / kSuper;                       // kSuper is a multiple of kChunk
        const int64_t blk_end = ((sb + 1) * kSuper < n) ? (sb + 1) * kSuper : n;
        uint32_t prev = (j0 == 0) ? sentinel : skeys[j0 - 1];
        for (int64_t j = j0; j < j1; ++j) {
            const uint32_t key = skeys[j];
            if (key == sentinel) break;          // sentinels sort last: nothing further in this chunk
            const bool starts_true = (j == 0) || key != prev;
            const bool head = starts_true || (j % kSuper == 0);
            prev = key;
            if (!head) continue;
            // extent of the piece: [j, e_end)
            int64_t e_end = j;
            for (int c = lane_c; c < cpr; c += lpr) {
                V acc = vzero<VEC>();
                int64_t e = j;
                auto trip = [&](auto width) -> bool {
                    constexpr int U = decltype(width)::value;
                    bool m[U];
                    typename Src::Val sv[U];
                    V v[U];
                    bool run = true;
                    if constexpr (WIDE) {
                        uint32_t kk[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) kk[u] = skeys[(e + u < blk_end) ? e + u : blk_end - 1];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            run = run && (e + u < blk_end) && (kk[u] == key);
                            m[u] = run;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            run = run && (e + u < blk_end) && (skeys[e + u] == key);
                            m[u] = run;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (m[u]) sv[u] = svals[e + u];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        v[u] = m[u] ? src.template load<VEC>(sv[u], cpr, c) : vzero<VEC>();
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (m[u]) {
                            vadd(acc, v[u]);
                            e_end = e + u + 1;
                        }
                    if (!m[U - 1]) return false;
                    e += U;
                    return true;
                };
                if constexpr (WIDE) {
                    while (trip(std::integral_constant<int, Src::kInFlight>{})) {
                    }
                } else {
                    while (trip(std::integral_constant<int, 4>{})) {
                    }
                }
                {
                    const bool ends_true = (e_end == n) || skeys[e_end] != key;
                    V* dst;
                    bool add;
                    if (starts_true && ends_true) {
                        dst = reinterpret_cast<V*>(grad_table) + (int64_t)key * cpr + c;
                        add = true;
                    } else if (!starts_true) {
                        dst = reinterpret_cast<V*>(head_piece) + sb * cpr + c;
                        add = false;
                    } else {
                        dst = reinterpret_cast<V*>(tail_piece) + sb * cpr + c;
                        add = false;
                    }
                    if (add) {
                        V cur = *dst;
                        vadd(cur, acc);
                        *dst = cur;
                    } else {
                        *dst = acc;
                    }
                }
            }
        }
    }
}

// Pass 1, balanced form: one group walks one block of SB consecutive sorted entries, whatever runs it holds, so
// every group has the same amount of work and long runs are summed by many groups at once.  Partial sums of runs
// that cross a block boundary go to head_piece / tail_piece exactly as in segment_sum_kernel (with kSuper = SB);
// runs that lie inside one block are added to their row directly (one writer per row).
template <int VEC, class Src, int SB>
__global__ __launch_bounds__(kBlock) void segment_sum_balanced_kernel(Src src, const uint32_t* __restrict__ skeys,
                                                                      const typename Src::Val* __restrict__ svals,
                                                                      int64_t n, int cpr, int lpr_log2,
                                                                      uint32_t sentinel, float* __restrict__ grad_table,
                                                                      float* __restrict__ head_piece,
                                                                      float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    constexpr int U = Src::kInFlight;
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
                } else if (!starts_true) {
                    reinterpret_cast<V*>(head_piece)[b * cpr + c] = acc;
                } else {
                    reinterpret_cast<V*>(tail_piece)[b * cpr + c] = acc;
                }
            };
            for (int64_t e = j0; e < j1; e += U) {
                uint32_t key[U];
                typename Src::Val sv[U];
                V v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = (e + u < j1) ? skeys[e + u] : sentinel;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (key[u] != sentinel) sv[u] = svals[e + u];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = (key[u] != sentinel) ? src.template load<VEC>(sv[u], cpr, c) : vzero<VEC>();
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
            }
            flush(cur != key_after);
        }
    }
}

// Pass 2: one group per block b of SB entries; acts only if a run starts inside b and continues past its end.
template <int VEC, int SB>
__global__ __launch_bounds__(kBlock) void segment_join_kernel(const uint32_t* __restrict__ skeys, int64_t n, int cpr,
                                                              int lpr_log2, uint32_t sentinel,
                                                              float* __restrict__ grad_table,
                                                              const float* __restrict__ head_piece,
                                                              const float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    constexpr int U = 8;   // head pieces fetched per trip (their "does the run go on" keys with them)
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int groups_per_block = kBlock >> lpr_log2;
    const int64_t n_sb = (n + SB - 1) / SB;
    for (int64_t b = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> lpr_log2); b < n_sb;
         b += (int64_t)gridDim.x * groups_per_block) {
        const int64_t end = (b + 1) * SB;
        if (end >= n) continue;                                   // last block: nothing continues
        const uint32_t key = skeys[end - 1];
        if (key == sentinel || skeys[end] != key) continue;       // no run crosses this boundary
        if (skeys[b * SB] == key && b > 0 && skeys[b * SB - 1] == key) continue;  // run started earlier
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
        }
    }
}

inline int sort_bits(int64_t n_rows) {
    int b = 1;
    while (((int64_t)1 << b) <= n_rows) ++b;  // sentinel == n_rows must be representable
    return b;
}

template <class Val>
inline size_t rocprim_sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    Val* v = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, p, p, v, v, (size_t)n, 0u, (unsigned)bits, hipStreamDefault);
    if (e != hipSuccess) return 0;
    return bytes;
}


inline size_t piece_bytes(int64_t n, int64_t d, int piece) {
    return hmx_align_up((size_t)hmx_ceil_div(n, piece) * (size_t)d * sizeof(float), 256);
}

template <class Src>
inline size_t workspace_bytes(int64_t n, int64_t n_rows, int64_t d) {
    if (n <= 0 || n_rows <= 0 || d <= 0) return 256;
    const size_t karr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    const size_t varr = hmx_align_up((size_t)n * sizeof(typename Src::Val), 256);
    return 2 * karr + 2 * varr + 2 * piece_bytes(n, d, Src::kPiece) +
           hmx_align_up(rocprim_sort_temp_bytes<typename Src::Val>(n, sort_bits(n_rows)), 256) + 256;
}

// table[ids[e]] += value(e) for e in [0, n).  `vec`: rows and table are 16-byte aligned and d % 4 == 0.
// sorted_keys_out (optional) receives the device pointer of the sorted keys inside the workspace (n entries,
// `sentinel_out` marks entries that did not contribute; they sort last).
template <class Src>
int sorted_segment_add(const Src& src, bool vec, const int64_t* ids, const int64_t* token_type, int64_t want_type,
                       int64_t n, int64_t d, float* table, int64_t n_rows, int64_t padding_idx, void* workspace,
                       size_t ws_bytes, hipStream_t s, const uint32_t** sorted_keys_out = nullptr,
                       uint32_t* sentinel_out = nullptr) {
    const int bits = sort_bits(n_rows);
    using Val = typename Src::Val;
    size_t temp_bytes = rocprim_sort_temp_bytes<Val>(n, bits);
    if (ws_bytes < workspace_bytes<Src>(n, n_rows, d) - 256) return HMX_EWORKSPACE;
    HmxCarver ws(workspace);
    uint32_t* keys = ws.take<uint32_t>(n);
    Val* vals = ws.take<Val>(n);
    uint32_t* skeys = ws.take<uint32_t>(n);
    Val* svals = ws.take<Val>(n);
    float* head_piece = ws.take<float>(piece_bytes(n, d, Src::kPiece) / sizeof(float));
    float* tail_piece = ws.take<float>(piece_bytes(n, d, Src::kPiece) / sizeof(float));
    void* temp = ws.take<char>(temp_bytes);
    const uint32_t sentinel = (uint32_t)n_rows;
    if (sorted_keys_out) *sorted_keys_out = skeys;
    if (sentinel_out) *sentinel_out = sentinel;

    const int grid1 = (int)std::min<int64_t>(hmx_ceil_div(n, kBlock), kMaxGrid);
    hipLaunchKernelGGL(make_keys_kernel<Src>, dim3(grid1), dim3(kBlock), 0, s, src, ids, token_type, want_type, n,
                       n_rows, padding_idx, sentinel, keys, vals);
    HMX_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, temp_bytes, keys, skeys, vals, svals, (size_t)n, 0u, (unsigned)bits, s) !=
        hipSuccess)
        return HMX_ELAUNCH;

    const int64_t cpr = vec ? d / 4 : d;
    const int l2 = lpr_log2_for(cpr);
    constexpr int SB = Src::kPiece;
    const int64_t n_sb = hmx_ceil_div(n, SB);
    const int64_t groups = Src::kBalanced ? n_sb : hmx_ceil_div(n, kChunk);
    const int grid3 = (int)std::min<int64_t>(hmx_ceil_div(groups, kBlock >> l2), (int64_t)kMaxGrid * 4);
    const int grid4 = (int)std::min<int64_t>(hmx_ceil_div(n_sb, kBlock >> l2), (int64_t)kMaxGrid);
    const bool long_runs = n >= 8 * n_rows;   // on average eight or more entries per table row
    if (vec) {
        if constexpr (Src::kBalanced)
            hipLaunchKernelGGL((segment_sum_balanced_kernel<4, Src, SB>), dim3(grid3), dim3(kBlock), 0, s, src, skeys,
                               svals, n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
        else
            if (long_runs)
                hipLaunchKernelGGL((segment_sum_kernel<4, Src, true>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals,
                                   n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
            else
                hipLaunchKernelGGL((segment_sum_kernel<4, Src, false>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals,
                                   n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
        hipLaunchKernelGGL((segment_join_kernel<4, SB>), dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2,
                           sentinel, table, head_piece, tail_piece);
    } else {
        if constexpr (Src::kBalanced)
            hipLaunchKernelGGL((segment_sum_balanced_kernel<1, Src, SB>), dim3(grid3), dim3(kBlock), 0, s, src, skeys,
                               svals, n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
        else
            if (long_runs)
                hipLaunchKernelGGL((segment_sum_kernel<1, Src, true>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals,
                                   n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
            else
                hipLaunchKernelGGL((segment_sum_kernel<1, Src, false>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals,
                                   n, (int)cpr, l2, sentinel, table, head_piece, tail_piece);
        hipLaunchKernelGGL((segment_join_kernel<1, SB>), dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2,
                           sentinel, table, head_piece, tail_piece);
    }
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

}  // namespace
}  // namespace hmx_seg
