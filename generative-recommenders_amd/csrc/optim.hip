// The optimizer end of the train step (main.py:709-713: clip_grad_norm_(1.0), AdamW(betas (0.9, 0.98), weight decay 0.1)
// step) over a list of tensors, as two passes over the gradients and one over the parameters and moments:
//   1. sum of squares of every gradient, one partial per chunk, added in a fixed order -> total norm and the clip factor
//      min(1, max_norm / (norm + 1e-6)) left on the device (no host round trip);
//   2. the AdamW update with the gradient multiplied by that factor on its way in.
// A step over the two embedding tables (1.28 GB of parameters) is a pure stream: 4 arrays read, 3 written.
#include <cmath>

#include "hmx_common.h"
#include "profile.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 16384;          // elements per workgroup
constexpr int kMaxTensors = 1023;      // chunk offsets of all tensors are staged in LDS

// index of the tensor that owns chunk c: the last i with chunk_start[i] <= c
__device__ __forceinline__ int owner_of(const int* cs, int n_tensors, int c) {
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cs[mid] <= c) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float block_sum(float v, float* wave_part /* [kThreads / 64] */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) s += wave_part[i];
    return s;
}

__global__ __launch_bounds__(kThreads) void grad_sumsq_kernel(const hmx_adamw_tensor* __restrict__ tensors,
                                                              const int* __restrict__ chunk_start, int n_tensors,
                                                              float* __restrict__ partial) {
    __shared__ int cs[kMaxTensors + 1];
    __shared__ float wave_part[kThreads / 64];
    for (int i = threadIdx.x; i <= n_tensors; i += kThreads) cs[i] = chunk_start[i];
    __syncthreads();
    const int c = blockIdx.x;
    const int ti = owner_of(cs, n_tensors, c);
    const hmx_adamw_tensor t = tensors[ti];
    const int64_t e0 = (int64_t)(c - cs[ti]) * kChunk;
    const int64_t n = (t.n - e0 < kChunk) ? t.n - e0 : kChunk;
    const float* g = t.g + e0;
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += kThreads) {
            const float4 v = reinterpret_cast<const float4*>(g)[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) s += g[i] * g[i];
    } else {
        for (int64_t i = threadIdx.x; i < n; i += kThreads) s += g[i] * g[i];
    }
    s = block_sum(s, wave_part);
    if (threadIdx.x == 0) partial[c] = s;
}

// one workgroup: partials added in index order per thread, then a fixed tree; out[0] = norm, out[1] = clip factor
__global__ __launch_bounds__(1024) void grad_norm_finish_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                                float* __restrict__ out) {
    __shared__ float part[1024 / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 1024 / 64; ++i) tot += part[i];
        const float norm = sqrtf(tot);
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;   // clip_grad_norm_'s factor
    }
}

struct AdamHyper {
    float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt;
};

// torch.optim.AdamW's update (its single-tensor form, in the same order of operations):
//   p *= 1 - lr wd;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamHyper& h) {
    p = p * (1.0f - h.lr * h.weight_decay);
    m = m + (g - m) * (1.0f - h.beta1);
    v = h.beta2 * v + (1.0f - h.beta2) * g * g;
    const float denom = sqrtf(v) / h.bias_correction2_sqrt + h.eps;
    p = p - (h.lr / h.bias_correction1) * (m / denom);
}

// Every element is read once and written once per step and nothing else in the step reads it soon: the streams
// bypass the caches' retention (nontemporal), which leaves L2 / Infinity Cache to whatever the next kernels need.
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float4 ldnt(const float* base, int64_t i) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stnt(float* base, int64_t i, const float4& v) {
    f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(base) + i);
}

__global__ __launch_bounds__(kThreads) void adamw_kernel(const hmx_adamw_tensor* __restrict__ tensors,
                                                         const int* __restrict__ chunk_start, int n_tensors,
                                                         AdamHyper h, const float* __restrict__ grad_scale,
                                                         int write_scaled_grads) {
    __shared__ int cs[kMaxTensors + 1];
    for (int i = threadIdx.x; i <= n_tensors; i += kThreads) cs[i] = chunk_start[i];
    __syncthreads();
    const int c = blockIdx.x;
    const int ti = owner_of(cs, n_tensors, c);
    const hmx_adamw_tensor t = tensors[ti];
    const int64_t e0 = (int64_t)(c - cs[ti]) * kChunk;
    const int64_t n = (t.n - e0 < kChunk) ? t.n - e0 : kChunk;
    float* p = t.p + e0;
    float* g = const_cast<float*>(t.g) + e0;
    float* m = t.m + e0;
    float* v = t.v + e0;
    const float gs = grad_scale ? *grad_scale : 1.0f;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    int64_t done = 0;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 2 * kThreads) {
            const int64_t j = i + kThreads;
            const bool two = j < n4;
            float4 p0 = ldnt(p, i), g0 = ldnt(g, i);
            float4 m0 = ldnt(m, i), v0 = ldnt(v, i);
            float4 p1 = p0, g1 = g0, m1 = m0, v1 = v0;
            if (two) {
                p1 = ldnt(p, j);
                g1 = ldnt(g, j);
                m1 = ldnt(m, j);
                v1 = ldnt(v, j);
            }
            g0.x *= gs; g0.y *= gs; g0.z *= gs; g0.w *= gs;
            adamw1(p0.x, g0.x, m0.x, v0.x, h);
            adamw1(p0.y, g0.y, m0.y, v0.y, h);
            adamw1(p0.z, g0.z, m0.z, v0.z, h);
            adamw1(p0.w, g0.w, m0.w, v0.w, h);
            stnt(p, i, p0);
            stnt(m, i, m0);
            stnt(v, i, v0);
            if (write_scaled_grads) stnt(g, i, g0);
            if (two) {
                g1.x *= gs; g1.y *= gs; g1.z *= gs; g1.w *= gs;
                adamw1(p1.x, g1.x, m1.x, v1.x, h);
                adamw1(p1.y, g1.y, m1.y, v1.y, h);
                adamw1(p1.z, g1.z, m1.z, v1.z, h);
                adamw1(p1.w, g1.w, m1.w, v1.w, h);
                stnt(p, j, p1);
                stnt(m, j, m1);
                stnt(v, j, v1);
                if (write_scaled_grads) stnt(g, j, g1);
            }
        }
        done = n4 << 2;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += kThreads) {
        float pp = p[i], gg = g[i] * gs, mm = m[i], vv = v[i];
        adamw1(pp, gg, mm, vv, h);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (write_scaled_grads) g[i] = gg;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Embedding tables: the same update, organised by ROWS.  A step's dense table gradient is zero outside the few
// hundred thousand rows the batch touched; with per-row flags (set by the scatter-add kernels) the gradient of an
// untouched row is not read (it is zero), a touched row's gradient is zeroed and its flag cleared once it has been
// used -- so the buffer is already the next step's zero-filled gradient -- and the bf16 scoring copy of the table
// (rows + largest row norm) is written from the registers that hold the updated parameters.
constexpr int kTableRows = 256;        // rows per workgroup
constexpr int kMaxTables = 8;
struct TableList {
    hmx_adamw_table t[kMaxTables];
    int block0[kMaxTables + 1];        // first workgroup of each table
    int n;
};

__device__ __forceinline__ int table_of(const TableList& tl, int b) {
    int i = 0;
    while (i + 1 < tl.n && tl.block0[i + 1] <= b) ++i;
    return i;
}

// partial[blockIdx.x] = sum of squares of the gradient rows of this workgroup's row block (flagged rows only)
__global__ __launch_bounds__(kThreads) void table_sumsq_kernel(TableList tl, float* __restrict__ partial) {
    __shared__ float wave_part[kThreads / 64];
    const int ti = table_of(tl, blockIdx.x);
    const hmx_adamw_table t = tl.t[ti];
    const int cpr = (int)(t.d / 4);
    int lpr = 1;
    while (lpr < cpr) lpr <<= 1;
    const int lane_c = threadIdx.x & (lpr - 1), rows_per_pass = kThreads / lpr;
    const int64_t r0 = (int64_t)(blockIdx.x - tl.block0[ti]) * kTableRows;
    float s = 0.f;
    for (int rr = threadIdx.x / lpr; rr < kTableRows; rr += rows_per_pass) {
        const int64_t r = r0 + rr;
        if (r >= t.n_rows || lane_c >= cpr) continue;
        if (t.row_flags != nullptr && t.row_flags[r] == 0) continue;
        const float4 v = ldnt(t.g + r * t.d, lane_c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = block_sum(s, wave_part);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(kThreads) void adamw_table_kernel(TableList tl, AdamHyper h, const float* __restrict__ grad_scale) {
    __shared__ float wave_max[kThreads / 64];
    const int ti = table_of(tl, blockIdx.x);
    const hmx_adamw_table t = tl.t[ti];
    const int cpr = (int)(t.d / 4);
    int lpr = 1;
    while (lpr < cpr) lpr <<= 1;
    const int lane_c = threadIdx.x & (lpr - 1), rows_per_pass = kThreads / lpr;
    const int64_t r0 = (int64_t)(blockIdx.x - tl.block0[ti]) * kTableRows;
    const float gs = grad_scale ? *grad_scale : 1.0f;
    float best = 0.f;
    // two rows per thread in flight
    for (int rr = threadIdx.x / lpr; rr < kTableRows; rr += 2 * rows_per_pass) {
        int64_t r[2] = {r0 + rr, r0 + rr + rows_per_pass};
        bool live[2], touched[2];
        float4 p4[2], g4[2], m4[2], v4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            live[u] = r[u] < t.n_rows && rr + u * rows_per_pass < kTableRows && lane_c < cpr;
            touched[u] = live[u] && (t.row_flags == nullptr || t.row_flags[r[u]] != 0);
            g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[u]) {
                p4[u] = ldnt(t.p + r[u] * t.d, lane_c);
                m4[u] = ldnt(t.m + r[u] * t.d, lane_c);
                v4[u] = ldnt(t.v + r[u] * t.d, lane_c);
                if (touched[u]) g4[u] = ldnt(t.g + r[u] * t.d, lane_c);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float n2 = 0.f;
            if (live[u]) {
                g4[u].x *= gs; g4[u].y *= gs; g4[u].z *= gs; g4[u].w *= gs;
                adamw1(p4[u].x, g4[u].x, m4[u].x, v4[u].x, h);
                adamw1(p4[u].y, g4[u].y, m4[u].y, v4[u].y, h);
                adamw1(p4[u].z, g4[u].z, m4[u].z, v4[u].z, h);
                adamw1(p4[u].w, g4[u].w, m4[u].w, v4[u].w, h);
                stnt(t.p + r[u] * t.d, lane_c, p4[u]);
                stnt(t.m + r[u] * t.d, lane_c, m4[u]);
                stnt(t.v + r[u] * t.d, lane_c, v4[u]);
                if (touched[u] && t.row_flags != nullptr) {      // used: back to the zero gradient, flag cleared
                    stnt(t.g + r[u] * t.d, lane_c, make_float4(0.f, 0.f, 0.f, 0.f));
                    if (lane_c == 0) t.row_flags[r[u]] = 0;
                }
                if (t.bf16_out != nullptr) {
                    uint2 o;
                    o.x = hmx_pack_bf16x2(p4[u].x, p4[u].y);
                    o.y = hmx_pack_bf16x2(p4[u].z, p4[u].w);
                    *reinterpret_cast<uint2*>(t.bf16_out + r[u] * t.bf16_ld + 4 * lane_c) = o;
                    n2 = p4[u].x * p4[u].x + p4[u].y * p4[u].y + p4[u].z * p4[u].z + p4[u].w * p4[u].w;
                }
            }
            else if (t.bf16_out != nullptr && r[u] < t.n_rows && rr + u * rows_per_pass < kTableRows && lane_c >= cpr &&
                     4 * lane_c < t.bf16_ld) {
                // a lane past the row's width: it writes the zero columns of the copy's padded row, so that the row
                // leaves as whole cache lines (120 of 128 bytes at d = 60 made every line a partial write)
                *reinterpret_cast<uint2*>(t.bf16_out + r[u] * t.bf16_ld + 4 * lane_c) = make_uint2(0u, 0u);
            }
            if (t.bf16_out != nullptr) {     // row norm: over the lpr lanes of the row (wave-uniform branch)
                for (int off = lpr >> 1; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
                best = fmaxf(best, n2);
            }
        }
    }
    if (t.bf16_out != nullptr && t.bf16_max_norm != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = wave_max[0];
#pragma unroll
            for (int i = 1; i < kThreads / 64; ++i) m = fmaxf(m, wave_max[i]);
            // slightly inflated so that the rounding of this reduction can never under-state a norm
            atomicMax(reinterpret_cast<unsigned*>(t.bf16_max_norm), __float_as_uint(sqrtf(m) * 1.0001f));
        }
    }
}

// host side: the by-value table list of a launch, or false when a table has an unsupported shape
bool make_table_list(const hmx_adamw_table* tables, int64_t n_tables, TableList& tl) {
    if (n_tables < 0 || n_tables > kMaxTables || (n_tables > 0 && !tables)) return false;
    tl.n = (int)n_tables;
    int b = 0;
    for (int i = 0; i < tl.n; ++i) {
        const hmx_adamw_table& t = tables[i];
        if (!t.p || !t.g || !t.m || !t.v || t.n_rows <= 0 || t.d <= 0 || t.d % 4 != 0 || t.d > 256) return false;
        if (((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
              reinterpret_cast<uintptr_t>(t.v)) & 15) != 0)
            return false;
        if (t.bf16_out != nullptr && (t.bf16_ld < t.d || t.bf16_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(t.bf16_out) & 7) != 0))
            return false;
        tl.t[i] = t;
        tl.block0[i] = b;
        b += (int)hmx_ceil_div(t.n_rows, kTableRows);
    }
    tl.block0[tl.n] = b;
    return true;
}

bool list_ok(const void* tensors, const void* chunk_start, int64_t n_tensors, int64_t total_chunks) {
    return tensors && chunk_start && n_tensors > 0 && n_tensors <= kMaxTensors && total_chunks > 0 &&
           total_chunks < ((int64_t)1 << 31);
}

}  // namespace

extern "C" int64_t hmx_adamw_chunk_elems(void) { return kChunk; }

extern "C" size_t hmx_grad_norm_workspace_bytes(int64_t total_chunks) {
    return hmx_align_up((size_t)(total_chunks > 0 ? total_chunks : 1) * sizeof(float), 256);
}

extern "C" int64_t hmx_adamw_table_blocks(int64_t n_rows) { return n_rows > 0 ? hmx_ceil_div(n_rows, kTableRows) : 0; }

extern "C" int hmx_grad_norm_f32(const hmx_adamw_tensor* tensors, const int32_t* chunk_start, int64_t n_tensors,
                                 int64_t total_chunks, const hmx_adamw_table* tables, int64_t n_tables, float max_norm,
                                 void* workspace, size_t workspace_bytes, float* norm_out, void* stream) {
    HMX_ENTER();
    TableList tl{};
    if (!make_table_list(tables, n_tables, tl) || !workspace || !norm_out) return HMX_EINVAL;
    if (n_tensors == 0) total_chunks = 0;
    if (n_tensors != 0 && !list_ok(tensors, chunk_start, n_tensors, total_chunks)) return HMX_EINVAL;
    const int64_t n_part = total_chunks + tl.block0[tl.n];
    if (n_part <= 0) return HMX_EINVAL;
    if (workspace_bytes < hmx_grad_norm_workspace_bytes(n_part)) return HMX_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = static_cast<float*>(workspace);
    void* span = hmx_profile_begin("optimizer", s);
    if (total_chunks > 0)
        hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)total_chunks), dim3(kThreads), 0, s, tensors, chunk_start,
                           (int)n_tensors, partial);
    if (tl.block0[tl.n] > 0)
        hipLaunchKernelGGL(table_sumsq_kernel, dim3((unsigned)tl.block0[tl.n]), dim3(kThreads), 0, s, tl, partial + total_chunks);
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(1024), 0, s, partial, (int)n_part, max_norm, norm_out);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_adamw_tables_step_f32(const hmx_adamw_table* tables, int64_t n_tables, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream) {
    HMX_ENTER();
    TableList tl{};
    if (!make_table_list(tables, n_tables, tl) || n_tables < 1 || step < 1) return HMX_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f)) return HMX_EINVAL;
    AdamHyper h{lr, beta1, beta2, eps, weight_decay, 0.f, 0.f};
    h.bias_correction1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    h.bias_correction2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < tl.n; ++i)
        if (tl.t[i].bf16_out != nullptr && tl.t[i].bf16_max_norm != nullptr &&
            hipMemsetAsync(tl.t[i].bf16_max_norm, 0, sizeof(float), s) != hipSuccess)
            return HMX_ELAUNCH;
    void* span = hmx_profile_begin("optimizer", s);
    hipLaunchKernelGGL(adamw_table_kernel, dim3((unsigned)tl.block0[tl.n]), dim3(kThreads), 0, s, tl, h, grad_scale);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_adamw_step_f32(const hmx_adamw_tensor* tensors, const int32_t* chunk_start, int64_t n_tensors,
                                  int64_t total_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int64_t step, const float* grad_scale, int write_scaled_grads, void* stream) {
    HMX_ENTER();
    if (!list_ok(tensors, chunk_start, n_tensors, total_chunks) || step < 1) return HMX_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f)) return HMX_EINVAL;
    AdamHyper h{lr, beta1, beta2, eps, weight_decay, 0.f, 0.f};
    h.bias_correction1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    h.bias_correction2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    hipStream_t s = (hipStream_t)stream;
    void* span = hmx_profile_begin("optimizer", s);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)total_chunks), dim3(kThreads), 0, s, tensors, chunk_start,
                       (int)n_tensors, h, grad_scale, write_scaled_grads);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
