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

bool list_ok(const void* tensors, const void* chunk_start, int64_t n_tensors, int64_t total_chunks) {
    return tensors && chunk_start && n_tensors > 0 && n_tensors <= kMaxTensors && total_chunks > 0 &&
           total_chunks < ((int64_t)1 << 31);
}

}  // namespace

extern "C" int64_t hmx_adamw_chunk_elems(void) { return kChunk; }

extern "C" size_t hmx_grad_norm_workspace_bytes(int64_t total_chunks) {
    return hmx_align_up((size_t)(total_chunks > 0 ? total_chunks : 1) * sizeof(float), 256);
}

extern "C" int hmx_grad_norm_f32(const hmx_adamw_tensor* tensors, const int32_t* chunk_start, int64_t n_tensors,
                                 int64_t total_chunks, float max_norm, void* workspace, size_t workspace_bytes,
                                 float* norm_out, void* stream) {
    HMX_ENTER();
    if (!list_ok(tensors, chunk_start, n_tensors, total_chunks) || !workspace || !norm_out) return HMX_EINVAL;
    if (workspace_bytes < hmx_grad_norm_workspace_bytes(total_chunks)) return HMX_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = static_cast<float*>(workspace);
    void* span = hmx_profile_begin("optimizer", s);
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)total_chunks), dim3(kThreads), 0, s, tensors, chunk_start,
                       (int)n_tensors, partial);
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(1024), 0, s, partial, (int)total_chunks, max_norm, norm_out);
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
