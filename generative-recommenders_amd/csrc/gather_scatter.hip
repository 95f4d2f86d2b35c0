// Row gather and deterministic scatter-add for fp32 embedding tables (HBM-bound).
//
// Layout: a table is (n_rows, d) row-major fp32; one row is d*4 bytes (256 B at d=64).
// A row is moved by a group of LPR = next_pow2(d/4) lanes, 16 B per lane, so a wave of
// 64 lanes moves 64/LPR whole rows per instruction and every access is a full 16-B vector
// inside one contiguous row segment.  d % 4 != 0 falls back to 4-B lanes.
#include <mutex>

#include "segment_sum.h"
#include "profile.h"

namespace {

using namespace hmx_seg;

constexpr int kUnroll = 4;

using f32x4_native = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void nt_store(const float4& v, float4* p) {
    f32x4_native x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4_native*>(p));
}
__device__ __forceinline__ void nt_store(const float& v, float* p) { __builtin_nontemporal_store(v, p); }

// exact GELU, x * Phi(x) (F.gelu default, embedding.py:644)
__device__ __forceinline__ float gelu1(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float4 vgelu(const float4& v) { return make_float4(gelu1(v.x), gelu1(v.y), gelu1(v.z), gelu1(v.w)); }
__device__ __forceinline__ float vgelu(const float& v) { return gelu1(v); }

// ROUTED: 0 = single table; 1 = item/user tables selected by token type (1 / 2, else zeros).
// GELU: the gathered row goes through GELU on its way out (candidate-embedding generation).
template <int VEC, bool ROUTED, bool GELU>
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
                if (r < n) nt_store(GELU ? vgelu(v[u]) : v[u], reinterpret_cast<V*>(out) + r * cpr + c);
            }
        }
    }
}

// Row gather with the copy rounded to bf16 on the way out (round to nearest even, v_cvt_pk_bf16_f32): the read
// side is the same random 16-byte-per-lane row fetch, the write stream is half as wide (8 bytes per lane).
// d % 4 == 0, 16-byte aligned table, 8-byte aligned output.
__global__ __launch_bounds__(kBlock) void gather_rows_bf16_kernel(const float* __restrict__ table, int64_t rows, int cpr,
                                                                  int lpr_log2, const int64_t* __restrict__ ids,
                                                                  int64_t n, uint2* __restrict__ out) {
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int rows_per_block = kBlock >> lpr_log2;
    const int64_t row_stride = (int64_t)gridDim.x * rows_per_block * kUnroll;
    for (int64_t base = (int64_t)blockIdx.x * rows_per_block * kUnroll + (threadIdx.x >> lpr_log2); base < n;
         base += row_stride) {
        const float4* src[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t r = base + (int64_t)u * rows_per_block;
            src[u] = nullptr;
            if (r < n) {
                const int64_t id = ids[r];
                if (id >= 0 && id < rows) src[u] = reinterpret_cast<const float4*>(table) + id * cpr;
            }
        }
        for (int c = lane_c; c < cpr; c += lpr) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) v[u] = src[u] ? src[u][c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t r = base + (int64_t)u * rows_per_block;
                if (r < n) {
                    using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
                    const u32x2 o = {hmx_pack_bf16x2(v[u].x, v[u].y), hmx_pack_bf16x2(v[u].z, v[u].w)};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(out) + r * cpr + c);
                }
            }
        }
    }
}

template <bool ROUTED, bool GELU = false>
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
        hipLaunchKernelGGL((gather_rows_kernel<4, ROUTED, GELU>), dim3(grid), dim3(kBlock), 0, s, t0, r0, t1, r1, (int)cpr,
                           l2, ids, tt, n, out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<1, ROUTED, GELU>), dim3(grid), dim3(kBlock), 0, s, t0, r0, t1, r1, (int)cpr,
                           l2, ids, tt, n, out);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

// flags[r] = 1 when row r of x (n_rows, d) holds a value != 0 (NaN counts), else 0: which rows of a dense table
// gradient a step touched.  A group of 2^lpr_log2 lanes owns a row (16-byte lanes when d % 4 == 0), four rows per
// group in flight, the group's verdict taken from one wave ballot; every lane stays in the loop (uniform trip count).
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void row_any_nonzero_kernel(const float* __restrict__ x, int64_t n_rows, int d,
                                                                 int lpr_log2, uint8_t* __restrict__ flags) {
    const int lpr = 1 << lpr_log2;
    const int lane = threadIdx.x & 63, l = lane & (lpr - 1), g0 = lane & ~(lpr - 1);
    const int rows_per_block = kBlock >> lpr_log2;
    const int64_t stride = (int64_t)gridDim.x * rows_per_block * kUnroll;
    const int64_t first = (int64_t)blockIdx.x * rows_per_block * kUnroll;
    const uint64_t gmask = lpr == 64 ? ~0ull : ((1ull << lpr) - 1);
    for (int64_t base = first; base < n_rows; base += stride) {          // block-uniform bound
        bool nz[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t r = base + (int64_t)u * rows_per_block + (threadIdx.x >> lpr_log2);
            nz[u] = false;
            if (r < n_rows) {
                if (VEC4) {
                    const float4* row = reinterpret_cast<const float4*>(x + r * d);
                    for (int c = l; c < d / 4; c += lpr) {
                        const float4 v = row[c];
                        nz[u] |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
                    }
                } else {
                    for (int c = l; c < d; c += lpr) nz[u] |= x[r * d + c] != 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t b = __ballot(nz[u]);
            const int64_t r = base + (int64_t)u * rows_per_block + (threadIdx.x >> lpr_log2);
            if (l == 0 && r < n_rows) flags[r] = ((b >> g0) & gmask) != 0 ? 1 : 0;
        }
    }
}

}  // namespace

extern "C" int hmx_gather_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n,
                                   float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0) return HMX_EINVAL;
    if (n > 0 && (!table || !ids || !out)) return HMX_EINVAL;
    return launch_gather<false>(table, n_rows, nullptr, 0, d, ids, nullptr, n, out, (hipStream_t)stream);
}

extern "C" int hmx_row_any_nonzero_f32(const float* x, int64_t n_rows, int64_t d, uint8_t* flags, void* stream) {
    HMX_ENTER();
    if (n_rows < 0 || d <= 0 || d >= ((int64_t)1 << 31)) return HMX_EINVAL;
    if (n_rows == 0) return HMX_OK;
    if (!x || !flags) return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = d % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
    const int l2 = lpr_log2_for(vec ? d / 4 : d);
    const int64_t rows_per_block = (kBlock >> l2) * kUnroll;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(n_rows, rows_per_block), kMaxGrid);
    void* span = hmx_profile_begin("row_any_nonzero", s);
    if (vec) hipLaunchKernelGGL(row_any_nonzero_kernel<true>, dim3(grid), dim3(kBlock), 0, s, x, n_rows, (int)d, l2, flags);
    else hipLaunchKernelGGL(row_any_nonzero_kernel<false>, dim3(grid), dim3(kBlock), 0, s, x, n_rows, (int)d, l2, flags);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gather_rows_bf16(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n,
                                    uint16_t* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0 || d % 4 != 0) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!table || !ids || !out || reinterpret_cast<uintptr_t>(table) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0)
        return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t cpr = d / 4;
    const int l2 = lpr_log2_for(cpr);
    const int64_t rows_per_block = (kBlock >> l2) * kUnroll;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(n, rows_per_block), kMaxGrid);
    void* span = hmx_profile_begin("gather_rows_bf16", s);
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(grid), dim3(kBlock), 0, s, table, n_rows, (int)cpr, l2, ids, n,
                       reinterpret_cast<uint2*>(out));
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gather_gelu_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n,
                                        float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0) return HMX_EINVAL;
    if (n > 0 && (!table || !ids || !out)) return HMX_EINVAL;
    return launch_gather<false, true>(table, n_rows, nullptr, 0, d, ids, nullptr, n, out, (hipStream_t)stream);
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

extern "C" size_t hmx_scatter_add_rows_workspace_bytes(int64_t n, int64_t n_rows, int64_t d) {
    return hmx_seg::workspace_bytes<hmx_seg::PlainRows>(n, n_rows, d);
}

static int scatter_add_impl(const float* grad_rows, const int64_t* ids, const int64_t* token_type, int64_t want_type,
                            int64_t n, int64_t d, float* grad_table, int64_t n_rows, int64_t padding_idx,
                            void* workspace, size_t workspace_bytes, hipStream_t s, uint8_t* row_flags = nullptr) {
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_rows) | reinterpret_cast<uintptr_t>(grad_table)) % 16 == 0);
    void* span = hmx_profile_begin("scatter_segment_sum", s);
    const int rc = hmx_seg::sorted_segment_add(hmx_seg::PlainRows{grad_rows}, vec, ids, token_type, want_type, n, d,
                                               grad_table, n_rows, padding_idx, workspace, workspace_bytes, s, nullptr,
                                               nullptr, row_flags);
    hmx_profile_end(span, s);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Scatter-add for callers that cannot allocate: the sort keys, sorted copies, run pieces and sort scratch live in a
// fixed array inside the library's code object instead of a caller-provided workspace.  Each GPU has its own copy
// of the array at its own address; `scratch` itself is only a host-side handle, so the device address is looked up
// once per device with hipGetSymbolAddress and that pointer is what the kernels receive.
constexpr size_t kScratchBytes = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

static int scratch_of_current_device(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return HMX_ELAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || ptr == nullptr) return HMX_ELAUNCH;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return HMX_OK;
}

extern "C" size_t hmx_scatter_add_rows_static_scratch_bytes(void) { return kScratchBytes; }

extern "C" int hmx_scatter_add_rows_f32_static(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                                               int64_t want_type, int64_t n, int64_t d, float* grad_table,
                                               int64_t n_rows, int64_t padding_idx, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0 || n >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31)) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!grad_rows || !ids || !grad_table) return HMX_EINVAL;
    if (hmx_seg::workspace_bytes<hmx_seg::PlainRows>(n, n_rows, d) > kScratchBytes) return HMX_ELIMIT;
    void* ptr = nullptr;
    const int rc = scratch_of_current_device(&ptr);
    if (rc != HMX_OK) return rc;
    // every word the kernels read from the scratch is written earlier in the same call (keys, values, their sorted
    // copies, the pieces a join reads), so nothing has to be cleared between calls
    return scatter_add_impl(grad_rows, ids, token_type, want_type, n, d, grad_table, n_rows, padding_idx, ptr,
                            kScratchBytes, (hipStream_t)stream);
}

extern "C" int hmx_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, const int64_t* token_type,
                                        int64_t want_type, int64_t n, int64_t d, float* grad_table, int64_t n_rows,
                                        int64_t padding_idx, uint8_t* row_flags, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0 || n >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31)) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!grad_rows || !ids || !grad_table || !workspace) return HMX_EINVAL;
    return scatter_add_impl(grad_rows, ids, token_type, want_type, n, d, grad_table, n_rows, padding_idx, workspace,
                            workspace_bytes, (hipStream_t)stream, row_flags);
}
