// Deterministic "sort, then sum runs" scatter-add shared by the embedding backward and the loss backward.
//
// out[key] += sum over the entries carrying that key of value(entry), where value(entry) is produced by a
// source functor (a materialised row, or coefficient x row).  The entries are sorted by key with a stable radix
// sort, so within a key they stay in entry order; see segment_sum_kernel for the fixed association order.
#pragma once
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "hmx_common.h"

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
    const float* rows;
    template <int VEC>
    __device__ __forceinline__ typename VecT<VEC>::type load(uint32_t pos, int cpr, int c) const {
        return reinterpret_cast<const typename VecT<VEC>::type*>(rows)[(int64_t)pos * cpr + c];
    }
};

__device__ __forceinline__ float4 vscale(float a, const float4& v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }
__device__ __forceinline__ float vscale(float a, const float& v) { return a * v; }

// value(entry) = coef[i, col0 + s] * rows[i] with entry = i * per_row + s: the transposed product of a sparse
// coefficient matrix with a dense row matrix, never materialised
struct WeightedRows {
    const float* rows;      // (n_src, d)
    const float* coef;      // (n_src, coef_stride)
    int per_row, coef_stride, col0;
    template <int VEC>
    __device__ __forceinline__ typename VecT<VEC>::type load(uint32_t pos, int cpr, int c) const {
        const uint32_t i = pos / (uint32_t)per_row, sidx = pos - i * (uint32_t)per_row;
        const float a = coef[(int64_t)i * coef_stride + col0 + sidx];
        return vscale(a, reinterpret_cast<const typename VecT<VEC>::type*>(rows)[(int64_t)i * cpr + c]);
    }
};

inline int lpr_log2_for(int64_t cpr) {
    int l = 0;
    while ((1 << l) < cpr && l < 6) ++l;
    return l;
}

// ---------------------------------------------------------------- scatter-add ----------
// Step 1: sort key = row id (or `sentinel` for entries that must not contribute), value = i.
__global__ __launch_bounds__(kBlock) void make_keys_kernel(const int64_t* __restrict__ ids,
                                                           const int64_t* __restrict__ token_type, int64_t want_type,
                                                           int64_t n, int64_t n_rows, int64_t padding_idx,
                                                           uint32_t sentinel, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t id = ids[i];
        bool ok = id >= 0 && id < n_rows && id != padding_idx;
        if (token_type != nullptr) ok = ok && token_type[i] == want_type;
        keys[i] = ok ? (uint32_t)id : sentinel;
        vals[i] = (uint32_t)i;
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

template <int VEC, class Src>
__global__ __launch_bounds__(kBlock) void segment_sum_kernel(Src src,
                                                             const uint32_t* __restrict__ skeys,
                                                             const uint32_t* __restrict__ svals, int64_t n, int cpr,
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
                while (true) {
                    bool m[4];
                    V v[4];
                    bool run = true;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        run = run && (e + u < blk_end) && (skeys[e + u] == key);
                        m[u] = run;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = m[u] ? src.template load<VEC>(svals[e + u], cpr, c) : vzero<VEC>();
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (m[u]) {
                            vadd(acc, v[u]);
                            e_end = e + u + 1;
                        }
                    if (!m[3]) break;
                    e += 4;
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

// Pass 2: one group per super-block b; acts only if a run starts inside b and continues past its end.
template <int VEC>
__global__ __launch_bounds__(kBlock) void segment_join_kernel(const uint32_t* __restrict__ skeys, int64_t n, int cpr,
                                                              int lpr_log2, uint32_t sentinel,
                                                              float* __restrict__ grad_table,
                                                              const float* __restrict__ head_piece,
                                                              const float* __restrict__ tail_piece) {
    using V = typename VecT<VEC>::type;
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int groups_per_block = kBlock >> lpr_log2;
    const int64_t n_sb = (n + kSuper - 1) / kSuper;
    for (int64_t b = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> lpr_log2); b < n_sb;
         b += (int64_t)gridDim.x * groups_per_block) {
        const int64_t end = (b + 1) * kSuper;
        if (end >= n) continue;                                   // last block: nothing continues
        const uint32_t key = skeys[end - 1];
        if (key == sentinel || skeys[end] != key) continue;       // no run crosses this boundary
        if (skeys[b * kSuper] == key && b > 0 && skeys[b * kSuper - 1] == key) continue;  // run started earlier
        for (int c = lane_c; c < cpr; c += lpr) {
            V acc = reinterpret_cast<const V*>(tail_piece)[b * cpr + c];
            int64_t cb = b + 1;
            while (true) {
                vadd(acc, reinterpret_cast<const V*>(head_piece)[cb * cpr + c]);
                const int64_t cend = (cb + 1) * kSuper;
                if (cend < n && skeys[cend - 1] == key && skeys[cend] == key) ++cb;
                else break;
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

inline size_t rocprim_sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)n, 0u, (unsigned)bits, hipStreamDefault);
    if (e != hipSuccess) return 0;
    return bytes;
}


inline size_t piece_bytes(int64_t n, int64_t d) {
    return hmx_align_up((size_t)hmx_ceil_div(n, kSuper) * (size_t)d * sizeof(float), 256);
}

inline size_t workspace_bytes(int64_t n, int64_t n_rows, int64_t d) {
    if (n <= 0 || n_rows <= 0 || d <= 0) return 256;
    const size_t arr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    return 4 * arr + 2 * piece_bytes(n, d) + hmx_align_up(rocprim_sort_temp_bytes(n, sort_bits(n_rows)), 256) + 256;
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
    size_t temp_bytes = rocprim_sort_temp_bytes(n, bits);
    const size_t arr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    if (ws_bytes < 4 * arr + 2 * piece_bytes(n, d) + hmx_align_up(temp_bytes, 256)) return HMX_EWORKSPACE;
    HmxCarver ws(workspace);
    uint32_t* keys = ws.take<uint32_t>(n);
    uint32_t* vals = ws.take<uint32_t>(n);
    uint32_t* skeys = ws.take<uint32_t>(n);
    uint32_t* svals = ws.take<uint32_t>(n);
    float* head_piece = ws.take<float>(piece_bytes(n, d) / sizeof(float));
    float* tail_piece = ws.take<float>(piece_bytes(n, d) / sizeof(float));
    void* temp = ws.take<char>(temp_bytes);
    const uint32_t sentinel = (uint32_t)n_rows;
    if (sorted_keys_out) *sorted_keys_out = skeys;
    if (sentinel_out) *sentinel_out = sentinel;

    const int grid1 = (int)std::min<int64_t>(hmx_ceil_div(n, kBlock), kMaxGrid);
    hipLaunchKernelGGL(make_keys_kernel, dim3(grid1), dim3(kBlock), 0, s, ids, token_type, want_type, n, n_rows,
                       padding_idx, sentinel, keys, vals);
    HMX_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, temp_bytes, keys, skeys, vals, svals, (size_t)n, 0u, (unsigned)bits, s) !=
        hipSuccess)
        return HMX_ELAUNCH;

    const int64_t cpr = vec ? d / 4 : d;
    const int l2 = lpr_log2_for(cpr);
    const int64_t groups = hmx_ceil_div(n, kChunk);
    const int grid3 = (int)std::min<int64_t>(hmx_ceil_div(groups, kBlock >> l2), (int64_t)kMaxGrid * 4);
    const int64_t n_sb = hmx_ceil_div(n, kSuper);
    const int grid4 = (int)std::min<int64_t>(hmx_ceil_div(n_sb, kBlock >> l2), (int64_t)kMaxGrid);
    if (vec) {
        hipLaunchKernelGGL((segment_sum_kernel<4, Src>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals, n, (int)cpr,
                           l2, sentinel, table, head_piece, tail_piece);
        hipLaunchKernelGGL(segment_join_kernel<4>, dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2, sentinel,
                           table, head_piece, tail_piece);
    } else {
        hipLaunchKernelGGL((segment_sum_kernel<1, Src>), dim3(grid3), dim3(kBlock), 0, s, src, skeys, svals, n, (int)cpr,
                           l2, sentinel, table, head_piece, tail_piece);
        hipLaunchKernelGGL(segment_join_kernel<1>, dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2, sentinel,
                           table, head_piece, tail_piece);
    }
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

}  // namespace
}  // namespace hmx_seg
