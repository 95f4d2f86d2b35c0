// Row operations of the feature-fusion paths of the embedding module (embedding.py:216-264, 486-611): bag sums of
// table rows (array features), the pointwise activations between the projections, the gate product of the gated
// fusion style and LayerNorm (modality encoders, cross-modal fusion).  All HBM-bound streaming kernels: 16-byte
// lanes, a row is handled by a group of next_pow2(d/4) lanes (bag sums) or by one wave (LayerNorm).
#include <algorithm>

#include "hmx_common.h"
#include "profile.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 16;

__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// out[r, :] = sum_{a < A} table[ids[r * A + a], :], added in the order of a; ids outside [0, rows) add nothing.
// VEC = 4: 16-byte columns (cpr = d / 4), VEC = 1: 4-byte columns (cpr = d).
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_sum_rows_kernel(const float* __restrict__ table, int64_t rows, int cpr,
                                                                 int lpr_log2, const int64_t* __restrict__ ids,
                                                                 int64_t n, int A, float* __restrict__ out) {
    const int lpr = 1 << lpr_log2;
    const int lane_c = threadIdx.x & (lpr - 1);
    const int rows_per_block = kBlock >> lpr_log2;
    for (int64_t r = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> lpr_log2); r < n;
         r += (int64_t)gridDim.x * rows_per_block) {
        const int64_t* bag = ids + r * A;
        for (int c = lane_c; c < cpr; c += lpr) {
            if (VEC == 4) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int a = 0;
                // two rows requested before the first is added (the adds stay in bag order)
                for (; a + 1 < A; a += 2) {
                    const int64_t i0 = bag[a], i1 = bag[a + 1];
                    const bool ok0 = i0 >= 0 && i0 < rows, ok1 = i1 >= 0 && i1 < rows;
                    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                    if (ok0) v0 = reinterpret_cast<const float4*>(table)[i0 * cpr + c];
                    if (ok1) v1 = reinterpret_cast<const float4*>(table)[i1 * cpr + c];
                    if (ok0) acc = f4add(acc, v0);
                    if (ok1) acc = f4add(acc, v1);
                }
                if (a < A) {
                    const int64_t i0 = bag[a];
                    if (i0 >= 0 && i0 < rows) acc = f4add(acc, reinterpret_cast<const float4*>(table)[i0 * cpr + c]);
                }
                reinterpret_cast<float4*>(out)[r * cpr + c] = acc;
            } else {
                float acc = 0.f;
                for (int a = 0; a < A; ++a) {
                    const int64_t i0 = bag[a];
                    if (i0 >= 0 && i0 < rows) acc += table[i0 * cpr + c];
                }
                out[r * cpr + c] = acc;
            }
        }
    }
}

// ---- pointwise --------------------------------------------------------------------------------
// mode 0: ReLU (embedding.py:606, :118, :134); mode 1: exact GELU x * Phi(x) (embedding.py:644, :106, :110, :127)
__device__ __forceinline__ float act1(float x, int mode) {
    return mode == 0 ? fmaxf(x, 0.f) : 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float act_grad1(float x, int mode) {
    if (mode == 0) return x > 0.f ? 1.f : 0.f;
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float sigmoid1(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename F>
__device__ __forceinline__ float4 map4(const float4& a, F f) { return make_float4(f(a.x), f(a.y), f(a.z), f(a.w)); }

__global__ __launch_bounds__(kBlock) void act_fwd_kernel(const float* __restrict__ x, int64_t n, int mode, int vec,
                                                         float* __restrict__ out) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    if (vec) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n / 4; i += step)
            reinterpret_cast<float4*>(out)[i] =
                map4(reinterpret_cast<const float4*>(x)[i], [mode](float v) { return act1(v, mode); });
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) out[i] = act1(x[i], mode);
    }
}

__global__ __launch_bounds__(kBlock) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         int64_t n, int mode, int vec, float* __restrict__ dx) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    if (vec) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n / 4; i += step) {
            const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(g)[i];
            reinterpret_cast<float4*>(dx)[i] =
                make_float4(b.x * act_grad1(a.x, mode), b.y * act_grad1(a.y, mode), b.z * act_grad1(a.z, mode),
                            b.w * act_grad1(a.w, mode));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) dx[i] = g[i] * act_grad1(x[i], mode);
    }
}

// out = p * sigmoid(q)   (embedding.py:254-255: a feature's projection times its learned gate)
__global__ __launch_bounds__(kBlock) void gate_fwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                          int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = p[i] * sigmoid1(q[i]);
}
__global__ __launch_bounds__(kBlock) void gate_bwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                          const float* __restrict__ g, int64_t n, float* __restrict__ dp,
                                                          float* __restrict__ dq) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float s = sigmoid1(q[i]);
        dp[i] = g[i] * s;
        dq[i] = g[i] * p[i] * s * (1.0f - s);
    }
}

// out = sigmoid(z) * a + (1 - sigmoid(z)) * l   (the gated mix of the multi-scale fusion tail, hydra3.py:536-541)
__global__ __launch_bounds__(kBlock) void gate_mix_fwd_kernel(const float* __restrict__ z, const float* __restrict__ a,
                                                              const float* __restrict__ l, int64_t n,
                                                              float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float s = sigmoid1(z[i]);
        out[i] = s * a[i] + (1.0f - s) * l[i];
    }
}
__global__ __launch_bounds__(kBlock) void gate_mix_bwd_kernel(const float* __restrict__ z, const float* __restrict__ a,
                                                              const float* __restrict__ l, const float* __restrict__ g,
                                                              int64_t n, float* __restrict__ dz, float* __restrict__ da,
                                                              float* __restrict__ dl) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float s = sigmoid1(z[i]);
        dz[i] = g[i] * (a[i] - l[i]) * s * (1.0f - s);
        da[i] = g[i] * s;
        dl[i] = g[i] * (1.0f - s);
    }
}

// ---- LayerNorm (nn.LayerNorm over the last dimension, biased variance, eps inside the root) -----------
constexpr int kLnMaxCols = 512;               // columns per row
constexpr int kLnPerLane = kLnMaxCols / 64;   // columns a lane owns: c = lane + 64 j

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kBlock) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, int64_t T, int d, float eps,
                                                               float* __restrict__ y, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < T; t += waves) {
        float v[kLnPerLane];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kLnPerLane; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < d ? x[t * d + c] : 0.f;
            s += v[j];
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < kLnPerLane; ++j) {
            const int c = lane + 64 * j;
            const float e = c < d ? v[j] - mean : 0.f;
            q += e * e;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int j = 0; j < kLnPerLane; ++j) {
            const int c = lane + 64 * j;
            if (c < d) y[t * d + c] = (v[j] - mean) * rstd * w[c] + b[c];
        }
        if (lane == 0) {
            mean_out[t] = mean;
            rstd_out[t] = rstd;
        }
    }
}

// ---- dropout multipliers: out[i] = 0 with probability p_drop, else 1 / (1 - p_drop) ---------------------
// Philox4x32-10 keyed by the caller's seed; counter = (block of four outputs, stream offset).  One counter yields four
// 32-bit words = four outputs, so a thread writes 16 bytes; the value of out[i] depends on (seed, offset, i) only.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

__global__ __launch_bounds__(kBlock) void dropout_multiplier_kernel(uint64_t seed, uint64_t offset, int64_t n, float keep,
                                                                    float scale, float* __restrict__ out) {
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < n4; b += (int64_t)gridDim.x * kBlock) {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f) < keep ? scale : 0.f;
        const int64_t i = b << 2;
        if (i + 4 <= n && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int j = 0; j < 4 && i + j < n; ++j) out[i + j] = v[j];
        }
    }
}

// ---- 1 / max(|x_t|, eps) per row: F.normalize(p=2) of wide rows, folded into the projection that follows
// as its input row scale.  One wave per row, 16-byte lanes when the row length allows, eight loads in flight.
__global__ __launch_bounds__(kBlock) void row_inv_norm_kernel(const float* __restrict__ x, int64_t T, int64_t d,
                                                              float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < T; t += waves) {
        const float* row = x + t * d;
        float s = 0.f;
        if ((d & 3) == 0) {
            const float4* r4 = reinterpret_cast<const float4*>(row);
            const int64_t n4 = d >> 2;
            int64_t c = lane;
            for (; c + 64 * 7 < n4; c += 64 * 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = r4[c + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
            }
            for (; c < n4; c += 64) {
                const float4 v = r4[c];
                s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            for (int64_t c = lane; c < d; c += 64) s += row[c] * row[c];
        }
        s = wave_sum(s);
        if (lane == 0) out[t] = 1.0f / fmaxf(sqrtf(s), eps);
    }
}

// dx per row; per-workgroup column partials of dw (first d columns of a partial row) and db (next d columns).
__global__ __launch_bounds__(kBlock) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               const float* __restrict__ g, int64_t T, int d,
                                                               float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[(kBlock / 64) * 2 * kLnMaxCols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    float pw[kLnPerLane], pb[kLnPerLane], wv[kLnPerLane];
#pragma unroll
    for (int j = 0; j < kLnPerLane; ++j) {
        const int c = lane + 64 * j;
        pw[j] = pb[j] = 0.f;
        wv[j] = c < d ? w[c] : 0.f;
    }
    for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + wave; t < T; t += waves) {
        const float mean = mean_in[t], rstd = rstd_in[t];
        float xh[kLnPerLane], gw[kLnPerLane];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < kLnPerLane; ++j) {
            const int c = lane + 64 * j;
            const float gv = c < d ? g[t * d + c] : 0.f;
            xh[j] = c < d ? (x[t * d + c] - mean) * rstd : 0.f;
            gw[j] = gv * wv[j];
            s1 += gw[j];
            s2 += gw[j] * xh[j];
            pw[j] += gv * xh[j];
            pb[j] += gv;
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int j = 0; j < kLnPerLane; ++j) {
            const int c = lane + 64 * j;
            if (c < d) dx[t * d + c] = rstd * (gw[j] - s1 - xh[j] * s2);
        }
    }
#pragma unroll
    for (int j = 0; j < kLnPerLane; ++j) {
        const int c = lane + 64 * j;
        red[(wave * 2 + 0) * kLnMaxCols + c] = pw[j];
        red[(wave * 2 + 1) * kLnMaxCols + c] = pb[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += kBlock) {
        const int which = c >= d, col = which ? c - d : c;
        float s = 0.f;
#pragma unroll
        for (int wv_i = 0; wv_i < kBlock / 64; ++wv_i) s += red[(wv_i * 2 + which) * kLnMaxCols + col];
        part[(int64_t)blockIdx.x * 2 * d + c] = s;
    }
}


// ---- multi-head attention core (nn.MultiheadAttention inside the cross-modal fusion, embedding.py:48-53, :76-78) ----
// q (B * Lq, d) and kv (B * Lk, 2 d) = [k | v] are the projected inputs, heads are column blocks of hd = d / H.
// One wave per (sequence, head, query): lanes own keys for the scores and the softmax, then feature columns for
// the weighted sum (weights parked in LDS).  drop (B, H, Lq, Lk), optional: dropout multiplier of the weights.
constexpr int kAttnMaxKeys = 4096;

__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                      const float* __restrict__ drop, int Lq, int Lk, int d, int H,
                                                      float* __restrict__ p_out, float* __restrict__ out) {
    __shared__ float w_sh[kAttnMaxKeys];
    const int lane = threadIdx.x;
    const int i = blockIdx.x % Lq, h = (blockIdx.x / Lq) % H, b = blockIdx.x / (Lq * H);
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* qh = q + ((int64_t)b * Lq + i) * d + h * hd;
    const float* kvb = kv + (int64_t)b * Lk * 2 * d;
    const int64_t prow = (((int64_t)b * H + h) * Lq + i) * Lk;
    float mx = -INFINITY;
    for (int l = lane; l < Lk; l += 64) {
        const float* kr = kvb + (int64_t)l * 2 * d + h * hd;
        float s = 0.f;
        for (int j = 0; j < hd; ++j) s = __builtin_fmaf(qh[j], kr[j], s);
        s *= scale;
        w_sh[l] = s;
        mx = fmaxf(mx, s);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int l = lane; l < Lk; l += 64) {
        const float e = expf(w_sh[l] - mx);
        w_sh[l] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int l = lane; l < Lk; l += 64) {
        const float pr = w_sh[l] * inv;
        p_out[prow + l] = pr;
        w_sh[l] = drop ? pr * drop[prow + l] : pr;
    }
    __syncthreads();
    for (int j = lane; j < hd; j += 64) {
        float acc = 0.f;
        for (int l = 0; l < Lk; ++l) acc = __builtin_fmaf(w_sh[l], kvb[(int64_t)l * 2 * d + d + h * hd + j], acc);
        out[((int64_t)b * Lq + i) * d + h * hd + j] = acc;
    }
}

// Per query: dw[l] = drop[l] <dout, v_l>, ds[l] = p[l] (dw[l] - sum_l p[l] dw[l]) / sqrt(hd), dq = sum_l ds[l] k_l.
// ds and the weights that multiplied v (p * drop) go to ds_out / wv_out (B, H, Lq, Lk) for the key-side kernel.
__global__ __launch_bounds__(64) void attn_bwd_q_kernel(const float* __restrict__ kv, const float* __restrict__ drop,
                                                        const float* __restrict__ p_in, const float* __restrict__ dout,
                                                        int Lq, int Lk, int d, int H, float* __restrict__ dq,
                                                        float* __restrict__ ds_out, float* __restrict__ wv_out) {
    __shared__ float ds_sh[kAttnMaxKeys];
    const int lane = threadIdx.x;
    const int i = blockIdx.x % Lq, h = (blockIdx.x / Lq) % H, b = blockIdx.x / (Lq * H);
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* doh = dout + ((int64_t)b * Lq + i) * d + h * hd;
    const float* kvb = kv + (int64_t)b * Lk * 2 * d;
    const int64_t prow = (((int64_t)b * H + h) * Lq + i) * Lk;
    float dot = 0.f;
    for (int l = lane; l < Lk; l += 64) {
        const float* vr = kvb + (int64_t)l * 2 * d + d + h * hd;
        float dw = 0.f;
        for (int j = 0; j < hd; ++j) dw = __builtin_fmaf(doh[j], vr[j], dw);
        const float m = drop ? drop[prow + l] : 1.0f;
        const float pr = p_in[prow + l];
        dw *= m;
        ds_sh[l] = dw;
        wv_out[prow + l] = pr * m;
        dot += pr * dw;
    }
    dot = wave_sum(dot);
    for (int l = lane; l < Lk; l += 64) {
        const float v = p_in[prow + l] * (ds_sh[l] - dot) * scale;
        ds_sh[l] = v;
        ds_out[prow + l] = v;
    }
    __syncthreads();
    for (int j = lane; j < hd; j += 64) {
        float acc = 0.f;
        for (int l = 0; l < Lk; ++l) acc = __builtin_fmaf(ds_sh[l], kvb[(int64_t)l * 2 * d + h * hd + j], acc);
        dq[((int64_t)b * Lq + i) * d + h * hd + j] = acc;
    }
}

// Per key: dk_l = sum_i ds[i, l] q_i, dv_l = sum_i wv[i, l] dout_i, queries added in order (one wave per
// (sequence, head, key), lanes own feature columns).
__global__ __launch_bounds__(64) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ dout,
                                                         const float* __restrict__ ds, const float* __restrict__ wv,
                                                         int Lq, int Lk, int d, int H, float* __restrict__ dkv) {
    const int lane = threadIdx.x;
    const int l = blockIdx.x % Lk, h = (blockIdx.x / Lk) % H, b = blockIdx.x / (Lk * H);
    const int hd = d / H;
    const int64_t pbase = ((int64_t)b * H + h) * Lq * Lk + l;
    for (int j = lane; j < hd; j += 64) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < Lq; ++i) {
            const int64_t row = ((int64_t)b * Lq + i) * d + h * hd + j;
            ak = __builtin_fmaf(ds[pbase + (int64_t)i * Lk], q[row], ak);
            av = __builtin_fmaf(wv[pbase + (int64_t)i * Lk], dout[row], av);
        }
        const int64_t o = ((int64_t)b * Lk + l) * 2 * d + h * hd + j;
        dkv[o] = ak;
        dkv[o + d] = av;
    }
}

int ln_blocks(int64_t T) { return (int)std::max<int64_t>(1, std::min<int64_t>(hmx_ceil_div(T, kBlock / 64), 2048)); }

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(hmx_ceil_div(n, kBlock), kMaxGrid)); }

bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace

extern "C" int hmx_gather_sum_rows_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* ids, int64_t n,
                                       int64_t bag, float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || d <= 0 || n_rows <= 0 || bag < 0 || bag > (1 << 20)) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!table || !out || (bag > 0 && !ids)) return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = d % 4 == 0 && aligned16(table, out);
    const int64_t cpr = vec ? d / 4 : d;
    int l2 = 0;
    while ((1 << l2) < cpr && l2 < 6) ++l2;
    const int64_t rows_per_block = kBlock >> l2;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(n, rows_per_block), kMaxGrid);
    void* span = hmx_profile_begin("gather_sum_rows", s);
    if (vec)
        hipLaunchKernelGGL(gather_sum_rows_kernel<4>, dim3(grid), dim3(kBlock), 0, s, table, n_rows, (int)cpr, l2, ids, n,
                           (int)bag, out);
    else
        hipLaunchKernelGGL(gather_sum_rows_kernel<1>, dim3(grid), dim3(kBlock), 0, s, table, n_rows, (int)cpr, l2, ids, n,
                           (int)bag, out);
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_act_fwd(const float* x, int64_t n, int mode, float* out, void* stream) {
    HMX_ENTER();
    if (n < 0 || mode < 0 || mode > 1) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!x || !out) return HMX_EINVAL;
    const int vec = n % 4 == 0 && aligned16(x, out);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(vec ? n / 4 : n)), dim3(kBlock), 0, (hipStream_t)stream, x, n, mode,
                       vec, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_act_bwd(const float* x, const float* g, int64_t n, int mode, float* dx, void* stream) {
    HMX_ENTER();
    if (n < 0 || mode < 0 || mode > 1) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!x || !g || !dx) return HMX_EINVAL;
    const int vec = n % 4 == 0 && aligned16(x, g, dx);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(vec ? n / 4 : n)), dim3(kBlock), 0, (hipStream_t)stream, x, g, n,
                       mode, vec, dx);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gate_fwd(const float* p, const float* q, int64_t n, float* out, void* stream) {
    HMX_ENTER();
    if (n < 0) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!p || !q || !out) return HMX_EINVAL;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, p, q, n, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gate_bwd(const float* p, const float* q, const float* g, int64_t n, float* dp, float* dq,
                            void* stream) {
    HMX_ENTER();
    if (n < 0) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!p || !q || !g || !dp || !dq) return HMX_EINVAL;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, p, q, g, n, dp, dq);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gate_mix_fwd(const float* z, const float* a, const float* l, int64_t n, float* out, void* stream) {
    HMX_ENTER();
    if (n < 0) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!z || !a || !l || !out) return HMX_EINVAL;
    hipLaunchKernelGGL(gate_mix_fwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, z, a, l, n, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_gate_mix_bwd(const float* z, const float* a, const float* l, const float* g, int64_t n, float* dz,
                                float* da, float* dl, void* stream) {
    HMX_ENTER();
    if (n < 0) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!z || !a || !l || !g || !dz || !da || !dl) return HMX_EINVAL;
    hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, z, a, l, g, n, dz, da, dl);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int64_t hmx_layernorm_bwd_blocks(int64_t T) { return ln_blocks(T); }

extern "C" int hmx_layernorm_fwd(const float* x, const float* weight, const float* bias, int64_t T, int64_t d, float eps,
                                 float* y, float* mean, float* rstd, void* stream) {
    HMX_ENTER();
    if (T < 0 || d <= 0 || d > kLnMaxCols) return HMX_EINVAL;
    if (T == 0) return HMX_OK;
    if (!x || !weight || !bias || !y || !mean || !rstd) return HMX_EINVAL;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(T, kBlock / 64), kMaxGrid);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, weight, bias, T, (int)d,
                       eps, y, mean, rstd);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_layernorm_bwd(const float* x, const float* weight, const float* mean, const float* rstd,
                                 const float* dy, int64_t T, int64_t d, float* dx, float* part, void* stream) {
    HMX_ENTER();
    if (T < 0 || d <= 0 || d > kLnMaxCols) return HMX_EINVAL;
    if (!part) return HMX_EINVAL;
    if (T > 0 && (!x || !weight || !mean || !rstd || !dy || !dx)) return HMX_EINVAL;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ln_blocks(T)), dim3(kBlock), 0, (hipStream_t)stream, x, weight, mean,
                       rstd, dy, T, (int)d, dx, part);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_dropout_multiplier_f32(uint64_t seed, uint64_t offset, int64_t n, float p_drop, float* out,
                                          void* stream) {
    HMX_ENTER();
    if (n < 0 || !(p_drop >= 0.f && p_drop <= 1.f)) return HMX_EINVAL;
    if (n == 0) return HMX_OK;
    if (!out) return HMX_EINVAL;
    const float keep = 1.0f - p_drop;
    const float scale = keep > 0.f ? 1.0f / keep : 0.f;
    hipLaunchKernelGGL(dropout_multiplier_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, seed,
                       offset, n, keep, scale, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_row_inv_norm_f32(const float* x, int64_t T, int64_t d, float eps, float* out, void* stream) {
    HMX_ENTER();
    if (T < 0 || d <= 0) return HMX_EINVAL;
    if (T == 0) return HMX_OK;
    if (!x || !out) return HMX_EINVAL;
    const int grid = (int)std::min<int64_t>(hmx_ceil_div(T, kBlock / 64), kMaxGrid);
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, T, d, eps, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

static bool attn_shape_ok(int64_t B, int64_t Lq, int64_t Lk, int64_t d, int64_t H) {
    return B >= 0 && Lq > 0 && Lk > 0 && Lk <= kAttnMaxKeys && d > 0 && H > 0 && d % H == 0 &&
           B * H * std::max(Lq, Lk) < ((int64_t)1 << 31);
}

extern "C" int hmx_attn_fwd(const float* q, const float* kv, const float* drop, int64_t B, int64_t Lq, int64_t Lk,
                            int64_t d, int64_t H, float* p, float* out, void* stream) {
    HMX_ENTER();
    if (!attn_shape_ok(B, Lq, Lk, d, H)) return HMX_EINVAL;
    if (B == 0) return HMX_OK;
    if (!q || !kv || !p || !out) return HMX_EINVAL;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * H * Lq)), dim3(64), 0, (hipStream_t)stream, q, kv, drop, (int)Lq,
                       (int)Lk, (int)d, (int)H, p, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_attn_bwd(const float* q, const float* kv, const float* drop, const float* p, const float* dout,
                            int64_t B, int64_t Lq, int64_t Lk, int64_t d, int64_t H, float* dq, float* dkv, float* ds,
                            float* wv, void* stream) {
    HMX_ENTER();
    if (!attn_shape_ok(B, Lq, Lk, d, H)) return HMX_EINVAL;
    if (B == 0) return HMX_OK;
    if (!q || !kv || !p || !dout || !dq || !dkv || !ds || !wv) return HMX_EINVAL;
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3((unsigned)(B * H * Lq)), dim3(64), 0, (hipStream_t)stream, kv, drop, p, dout,
                       (int)Lq, (int)Lk, (int)d, (int)H, dq, ds, wv);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3((unsigned)(B * H * Lk)), dim3(64), 0, (hipStream_t)stream, q, dout, ds, wv,
                       (int)Lq, (int)Lk, (int)d, (int)H, dkv);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
