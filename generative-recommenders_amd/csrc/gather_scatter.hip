// Row gather and deterministic scatter-add for fp32 embedding tables (HBM-bound).
//
// Layout: a table is (n_rows, d) row-major fp32; one row is d*4 bytes (256 B at d=64).
// A row is moved by a group of LPR = next_pow2(d/4) lanes, 16 B per lane, so a wave of
// 64 lanes moves 64/LPR whole rows per instruction and every access is a full 16-B vector
// inside one contiguous row segment.  d % 4 != 0 falls back to 4-B lanes.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "hmx_common.h"
#include "profile.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;
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

using f32x4_native = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void nt_store(const float4& v, float4* p) {
    f32x4_native x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4_native*>(p));
}
__device__ __forceinline__ void nt_store(const float& v, float* p) { __builtin_nontemporal_store(v, p); }

// ROUTED: 0 = single table; 1 = item/user tables selected by token type (1 / 2, else zeros).
template <int VEC, bool ROUTED>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ table0, int64_t rows0,
                                                             const float* __restrict__ table1, int64_t rows1,
                                                             int cpr, int lpr_log2,
                                                             const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ token_type, int64_t n,
                                                             float* __restrict__ out) {
    using V = typename VecT<VEC>::type;
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int rows_per_block = kBlock >> lpr_log2;
    const int64_t row_stride = (int64_t)gridDim.x * rows_per_block * kUnroll;
    for (int64_t base = (int64_t)blockIdx.x * rows_per_block * kUnroll + (threadIdx.x >> lpr_log2); base < n;
         base += row_stride) {
        const V* src[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t r = base + (int64_t)u * rows_per_block;
            src[u] = nullptr;
            if (r < n) {
                const int64_t id = ids[r];
                if (ROUTED) {
                    const int64_t tt = token_type[r];
                    if (tt == 1 && id >= 0 && id < rows0) src[u] = reinterpret_cast<const V*>(table0) + id * cpr;
                    else if (tt == 2 && id >= 0 && id < rows1) src[u] = reinterpret_cast<const V*>(table1) + id * cpr;
                } else if (id >= 0 && id < rows0) {
                    src[u] = reinterpret_cast<const V*>(table0) + id * cpr;
                }
            }
        }
        for (int c = lane_c; c < cpr; c += lpr) {
            V v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) v[u] = src[u] ? src[u][c] : vzero<VEC>();
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t r = base + (int64_t)u * rows_per_block;
                // streaming store: the gathered copy is not read again by this kernel, so it should not evict
                // table rows (which repeat) from L2 / Infinity Cache
                if (r < n) nt_store(v[u], reinterpret_cast<V*>(out) + r * cpr + c);
            }
        }
    }
}

int lpr_log2_for(int64_t cpr) {
    int l = 0;
    while ((1 << l) < cpr && l < 6) ++l;
    return l;
}

template <bool ROUTED>
int launch_gather(const float* t0, int64_t r0, const float* t1, int64_t r1, int64_t d, const int64_t* ids,
                  const int64_t* tt, int64_t n, float* out, hipStream_t s) {
    if (n == 0) return HMX_OK;
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(t0) | reinterpret_cast<uintptr_t>(t1) |
                                       reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int64_t cpr = vec ? d / 4 : d;
    const int l2 = lpr_log2_for(cpr);
    const int64_t rows_per_block = (kBlock >> l2) * kUnroll;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(n, rows_per_block), kMaxGrid);
    void* span = hmx_profile_begin("gather_rows", s);
    if (vec)
        hipLaunchKernelGGL((gather_rows_kernel<4, ROUTED>), dim3(grid), dim3(kBlock), 0, s, t0, r0, t1, r1, (int)cpr,
                           l2, ids, tt, n, out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<1, ROUTED>), dim3(grid), dim3(kBlock), 0, s, t0, r0, t1, r1, (int)cpr,
                           l2, ids, tt, n, out);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
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

template <int VEC>
__global__ __launch_bounds__(kBlock) void segment_sum_kernel(const float* __restrict__ grad_rows,
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
    const V* rows = reinterpret_cast<const V*>(grad_rows);
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
                        v[u] = m[u] ? rows[(int64_t)svals[e + u] * cpr + c] : vzero<VEC>();
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

int sort_bits(int64_t n_rows) {
    int b = 1;
    while (((int64_t)1 << b) <= n_rows) ++b;  // sentinel == n_rows must be representable
    return b;
}

size_t rocprim_sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)n, 0u, (unsigned)bits, hipStreamDefault);
    if (e != hipSuccess) return 0;
    return bytes;
}

}  // namespace

extern "C" int hmx_gather_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n,
                                   float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0) return HMX_EINVAL;
    if (n > 0 && (!table || !ids || !out)) return HMX_EINVAL;
    return launch_gather<false>(table, n_rows, nullptr, 0, d, ids, nullptr, n, out, (hipStream_t)stream);
}

extern "C" int hmx_gather_routed_f32(const float* item_table, int64_t item_rows, const float* user_table,
                                     int64_t user_rows, int64_t d, const int64_t* ids, const int64_t* token_type,
                                     int64_t n, float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || item_rows <= 0 || user_rows <= 0) return HMX_EINVAL;
    if (n > 0 && (!item_table || !user_table || !ids || !token_type || !out)) return HMX_EINVAL;
    return launch_gather<true>(item_table, item_rows, user_table, user_rows, d, ids, token_type, n, out,
                               (hipStream_t)stream);
}

static size_t piece_bytes(int64_t n, int64_t d) {
    return hmx_align_up((size_t)hmx_ceil_div(n, kSuper) * (size_t)d * sizeof(float), 256);
}

extern "C" size_t hmx_scatter_add_rows_workspace_bytes(int64_t n, int64_t n_rows, int64_t d) {
    if (n <= 0 || n_rows <= 0 || d <= 0) return 256;
    const size_t arr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    return 4 * arr + 2 * piece_bytes(n, d) + hmx_align_up(rocprim_sort_temp_bytes(n, sort_bits(n_rows)), 256) + 256;
}

extern "C" int hmx_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                                        int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                                        int64_t padding_idx, void* workspace, size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0 || n >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31)) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!grad_rows || !ids || !grad_table || !workspace) return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int bits = sort_bits(n_rows);
    size_t temp_bytes = rocprim_sort_temp_bytes(n, bits);
    const size_t arr = hmx_align_up((size_t)n * sizeof(uint32_t), 256);
    if (workspace_bytes < 4 * arr + 2 * piece_bytes(n, d) + hmx_align_up(temp_bytes, 256)) return HMX_EWORKSPACE;
    HmxCarver ws(workspace);
    uint32_t* keys = ws.take<uint32_t>(n);
    uint32_t* vals = ws.take<uint32_t>(n);
    uint32_t* skeys = ws.take<uint32_t>(n);
    uint32_t* svals = ws.take<uint32_t>(n);
    float* head_piece = ws.take<float>(piece_bytes(n, d) / sizeof(float));
    float* tail_piece = ws.take<float>(piece_bytes(n, d) / sizeof(float));
    void* temp = ws.take<char>(temp_bytes);
    const uint32_t sentinel = (uint32_t)n_rows;

    const int grid1 = (int)std::min<int64_t>(hmx_ceil_div(n, kBlock), kMaxGrid);
    hipLaunchKernelGGL(make_keys_kernel, dim3(grid1), dim3(kBlock), 0, s, ids, token_type, want_type, n, n_rows,
                       padding_idx, sentinel, keys, vals);
    HMX_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, temp_bytes, keys, skeys, vals, svals, (size_t)n, 0u, (unsigned)bits, s) !=
        hipSuccess)
        return HMX_ELAUNCH;

    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_rows) | reinterpret_cast<uintptr_t>(grad_table)) % 16 == 0);
    const int64_t cpr = vec ? d / 4 : d;
    const int l2 = lpr_log2_for(cpr);
    const int64_t groups = hmx_ceil_div(n, kChunk);
    const int grid3 = (int)std::min<int64_t>(hmx_ceil_div(groups, kBlock >> l2), (int64_t)kMaxGrid * 4);
    void* span = hmx_profile_begin("scatter_segment_sum", s);
    const int64_t n_sb = hmx_ceil_div(n, kSuper);
    const int grid4 = (int)std::min<int64_t>(hmx_ceil_div(n_sb, kBlock >> l2), (int64_t)kMaxGrid);
    if (vec) {
        hipLaunchKernelGGL(segment_sum_kernel<4>, dim3(grid3), dim3(kBlock), 0, s, grad_rows, skeys, svals, n, (int)cpr,
                           l2, sentinel, grad_table, head_piece, tail_piece);
        hipLaunchKernelGGL(segment_join_kernel<4>, dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2, sentinel,
                           grad_table, head_piece, tail_piece);
    } else {
        hipLaunchKernelGGL(segment_sum_kernel<1>, dim3(grid3), dim3(kBlock), 0, s, grad_rows, skeys, svals, n, (int)cpr,
                           l2, sentinel, grad_table, head_piece, tail_piece);
        hipLaunchKernelGGL(segment_join_kernel<1>, dim3(grid4), dim3(kBlock), 0, s, skeys, n, (int)cpr, l2, sentinel,
                           grad_table, head_piece, tail_piece);
    }
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
