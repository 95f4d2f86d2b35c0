// Kernels of the Hydra sequence encoder (forward and backward).
//
// Activations are (T = B*L tokens, C features) row-major fp32 in HBM.  Dense projections run on
// bf16 MFMA (32x32x16, fp32 accumulate) with the weight panel staged once per workgroup in LDS;
// weight gradients run on fp32 MFMA (32x32x2), whose operands load coalesced with the token
// index on the MFMA K axis.  The selective scan keeps its (N = 16)-wide state per channel in
// registers and walks the sequence; everything pointwise (SiLU, RoPE, gates, conv, RMSNorm) is
// fused into a handful of streaming kernels.
#include <algorithm>
#include <atomic>

#include <mutex>

#include "hmx_common.h"
#include "profile.h"

// The library is built with -ffp-contract=off because the gather / scatter / scoring kernels promise bit-exact sums in
// a stated order.  The encoder promises a tolerance instead (its projections run on bf16 MFMA), and its scan kernels
// are bound by vector-instruction issue: here a multiply followed by an add may fuse into one fma.
#pragma clang fp contract(fast)

namespace {

using short8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kN = 16;   // SSM state size   (d_state, hydra3.py:593)
constexpr int kR = 16;   // dt rank          (dt_rank, hydra3.py:591)
constexpr int kP = kR + 2 * kN;
constexpr int kMaxGrid = 4096;

// 1 / (1 + e^-x) with the hardware reciprocal (1 ulp) instead of an IEEE division (ten instructions)
__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_(float x) { return x * sigmoid_(x); }
__device__ __forceinline__ float silu_grad_(float x) {
    const float s = sigmoid_(x);
    return s * (1.f + x * (1.f - s));
}
// log(1 + e^x) = max(x, 0) + log(1 + e^-|x|): the argument of the logarithm stays in (1, 2], where the hardware
// logarithm is accurate to an ulp of the result; for x > 20 the sum is x in fp32 (the reference's threshold form).
__device__ __forceinline__ float softplus_(float x) {
    return x > 20.f ? x : fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x)));
}

template <class... P>
inline bool aligned16(const P*... p) {
    return ((reinterpret_cast<uintptr_t>(p) | ...) & 15) == 0;   // null pointers count as aligned
}
inline int grid_for(int64_t work, int per_block) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(hmx_ceil_div(work, per_block), kMaxGrid));
}

// ---------------------------------------------------------------------------- RMSNorm
// y = x / sqrt(mean(x^2) + eps) * scale      (hydra3.py:221-229; eps inside the root)
// Sixteen lanes per row; VEC = 4: lane l owns columns 4 l .. 4 l + 3 (+ 64 j), read and written 16 bytes at a time
// (d % 4 == 0, 16-byte aligned rows); VEC = 1: columns l + 16 i.
template <int VEC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          int64_t T, int d, float eps, float* __restrict__ y,
                                                          float* __restrict__ rstd) {
    const int l = threadIdx.x & 15;
    for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < hmx_ceil_div(T, 16) * 16;
         row += (int64_t)gridDim.x * 16) {
        const bool ok = row < T;
        float ss = 0.f;
        float4 xv[4];
        if (ok) {
            if constexpr (VEC == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c = 4 * l + 64 * jj;
                    xv[jj] = c < d ? *reinterpret_cast<const float4*>(x + row * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    ss += xv[jj].x * xv[jj].x + xv[jj].y * xv[jj].y + xv[jj].z * xv[jj].z + xv[jj].w * xv[jj].w;
                }
            } else {
                for (int c = l; c < d; c += 16) {
                    const float v = x[row * d + c];
                    ss += v * v;
                }
            }
        }
        for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float r = 1.f / sqrtf(ss / (float)d + eps);
        if (ok) {
            if constexpr (VEC == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c = 4 * l + 64 * jj;
                    if (c < d) {
                        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
                        *reinterpret_cast<float4*>(y + row * d + c) =
                            make_float4(xv[jj].x * r * sc.x, xv[jj].y * r * sc.y, xv[jj].z * r * sc.z, xv[jj].w * r * sc.w);
                    }
                }
            } else {
                for (int c = l; c < d; c += 16) y[row * d + c] = x[row * d + c] * r * scale[c];
            }
            if (l == 0 && rstd) rstd[row] = r;
        }
    }
}

// dx_j = r * (s_j dy_j - x_j r^2 mean_i(dy_i s_i x_i));  dscale_j = sum_t dy_j x_j r  (per-block partials)
template <int VEC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ rstd, const float* __restrict__ dy,
                                                          int64_t T, int d, float* __restrict__ dx,
                                                          float* __restrict__ dscale_part) {
    __shared__ float red[16][257];
    const int l = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float ds[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ds[i] = 0.f;
    // column of this lane's i-th element
    auto col = [&](int i) { return VEC == 4 ? 4 * l + (i & 3) + 64 * (i >> 2) : l + 16 * i; };
    for (int64_t row = (int64_t)blockIdx.x * 16 + slot; row < hmx_ceil_div(T, 16) * 16; row += (int64_t)gridDim.x * 16) {
        const bool ok = row < T;
        float dot = 0.f;
        float xr[16], gr[16], sr[16];
        if (ok) {
            if constexpr (VEC == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c = 4 * l + 64 * jj;
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a, sc = a;
                    if (c < d) {
                        a = *reinterpret_cast<const float4*>(x + row * d + c);
                        g = *reinterpret_cast<const float4*>(dy + row * d + c);
                        sc = *reinterpret_cast<const float4*>(scale + c);
                    }
                    xr[4 * jj] = a.x; xr[4 * jj + 1] = a.y; xr[4 * jj + 2] = a.z; xr[4 * jj + 3] = a.w;
                    gr[4 * jj] = g.x; gr[4 * jj + 1] = g.y; gr[4 * jj + 2] = g.z; gr[4 * jj + 3] = g.w;
                    sr[4 * jj] = sc.x; sr[4 * jj + 1] = sc.y; sr[4 * jj + 2] = sc.z; sr[4 * jj + 3] = sc.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = l + 16 * i;
                    xr[i] = c < d ? x[row * d + c] : 0.f;
                    gr[i] = c < d ? dy[row * d + c] : 0.f;
                    sr[i] = c < d ? scale[c] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dot += gr[i] * sr[i] * xr[i];
        }
        for (int off = 8; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        if (ok) {
            const float r = rstd[row];
            const float coef = dot * r * r / (float)d;
            float o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[i] = r * (sr[i] * gr[i] - xr[i] * coef);
                ds[i] += gr[i] * xr[i] * r;
            }
            if constexpr (VEC == 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c = 4 * l + 64 * jj;
                    if (c < d)
                        *reinterpret_cast<float4*>(dx + row * d + c) = make_float4(o[4 * jj], o[4 * jj + 1], o[4 * jj + 2], o[4 * jj + 3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (l + 16 * i < d) dx[row * d + l + 16 * i] = o[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = col(i);
        if (c < d) red[slot][c] = ds[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][c];
        dscale_part[(int64_t)blockIdx.x * d + c] = s;
    }
}

// ---------------------------------------------------------------------------- column sums
// out[c] = sum_r P[r * ld + c], rows summed in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ P, int64_t rows, int64_t cols, int64_t ld,
                                                      float* __restrict__ out) {
    // sixteen waves share the rows of one 64-column block, eight independent loads in flight per lane
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (c < cols) {
        int64_t r = w;
        for (; r + 7 * 16 < rows; r += 128) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += P[(r + 16 * k) * ld + c];
        }
        for (; r < rows; r += 16) s[0] += P[r * ld + c];
    }
    red[w][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (w == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        out[c] = t;
    }
}

// Two-stage form for tall inputs: block (cb, r) sums the rows of chunk r for 64 columns into part[r][c]; the second
// stage adds the chunks in order.  Same fixed order on every run.
__global__ __launch_bounds__(256) void colsum_chunks_kernel(const float* __restrict__ P, int64_t rows, int64_t cols,
                                                            int64_t ld, int64_t rows_per_chunk, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int64_t r = r0 + w;
        for (; r + 12 < r1; r += 16) {   // four independent loads in flight per lane
            s0 += P[r * ld + c];
            s1 += P[(r + 4) * ld + c];
            s2 += P[(r + 8) * ld + c];
            s3 += P[(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) s0 += P[r * ld + c];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < cols) part[(int64_t)blockIdx.y * cols + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
// 64 columns per workgroup; sixteen waves take every sixteenth chunk each and the first adds their sums in wave order
// (a narrow matrix has up to 1024 chunks: one wave walking them alone took 12 us)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, int n_chunks, int64_t cols,
                                                             float* __restrict__ out) {
    __shared__ float sums[16][64];
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 64 + l;
    float s = 0.f;
    if (c < cols)
        for (int r = g; r < n_chunks; r += 16) s += part[(int64_t)r * cols + c];
    sums[g][l] = s;
    __syncthreads();
    if (g == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sums[i][l];
        out[c] = t;
    }
}
// chunks of rows so that about a thousand blocks run; none when the input is short
int colsum_chunks(int64_t rows, int64_t cols) {
    if (rows <= 2048) return 1;   // at most 512 KB per 64-column block: one launch beats two
    const int64_t col_blocks = hmx_ceil_div(cols, 64);
    const int64_t want = std::max<int64_t>(1, 1024 / col_blocks);
    return (int)std::min<int64_t>(want, std::max<int64_t>(1, rows / 64));
}

// ---------------------------------------------------------------------------- linear (bf16 MFMA)
constexpr int kLinearMaxHeads = 8;
struct LinearParams {
    const float* X;
    int64_t ldx;
    const float* W;          // viewed as Wv(n, k) = W[n * w_rs + k * w_cs]
    int64_t w_rs, w_cs;
    const float* bias;       // (N) or null
    int64_t T;
    int N, K, n_chunk;
    float* Y;
    int64_t ldy;
    const float* R1;         // residual 1: Y += c1 * R1
    int64_t ldr1;
    float c1;
    const float* R2;
    int64_t ldr2;
    float c2;
    const float* in_rowmask;   // (T) or null: X row scaled before the product
    const float* out_rowmask;  // (T) or null: Y row scaled after everything else
    int accumulate;            // Y = Y_old + result
    // block-diagonal form (n_heads > 0): head h = blockIdx.z multiplies columns [h * x_hs, h * x_hs + K) of X by its
    // own weight Wh[h] into columns [h * y_hs, h * y_hs + N) of Y -- all heads of a projection in one launch
    int n_heads;
    int64_t x_hs, y_hs;
    const float* Wh[kLinearMaxHeads];
};

// SPLIT = false: one bf16 MFMA per product (inputs rounded to bf16, relative error ~2^-9 per operand).
// SPLIT = true : every operand is split x = hi + lo (two bf16 numbers, 16 significant bits together) and the product
//                is accumulated as hi*hi + lo*hi + hi*lo on the same MFMA: fp32-class accuracy at three MFMAs.
template <int KS, bool SPLIT>
__global__ __launch_bounds__(256, KS <= 4 ? 4 : 2) void linear_kernel(LinearParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = KS * 16 + 8;  // bf16 elements per LDS row (+8 keeps ds_read_b128 conflict-free)
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    uint16_t* wl_lo = wl + (size_t)p.n_chunk * KP;
    const float* __restrict__ Wm = p.W;
    const float* __restrict__ Xm = p.X;
    float* __restrict__ Ym = p.Y;
    if (p.n_heads > 0) {
        const int hd = blockIdx.z;
        Wm = p.Wh[hd];
        Xm += hd * p.x_hs;
        Ym += hd * p.y_hs;
    }
    const int n_base = blockIdx.y * p.n_chunk;
    const int nc = min(p.n_chunk, p.N - n_base);
    const int npad = (nc + 31) / 32 * 32;
    // Weight panel -> bf16 (hi / lo) in LDS.  The reads of a trip are all issued before the first is converted: one
    // read per trip made the panel a chain of 32 dependent cache round trips, longer than the token work of the
    // workgroup (K = 128 -> 48 over 102,400 tokens: 45 -> 38 us).  Forward weights (contiguous in k) come 16 bytes
    // per lane; any other view element by element; eight reads in flight either way.
    const int total = npad * KS * 16;
    if (p.w_cs == 1 && (p.w_rs & 3) == 0 && (p.K & 3) == 0 && (reinterpret_cast<uintptr_t>(Wm) & 15) == 0) {
        constexpr int kBatch = 8;
        for (int base = threadIdx.x; base < total / 4; base += 256 * kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = base + u * 256;
                const int n = idx / (KS * 4), k = (idx % (KS * 4)) * 4;
                v[u] = (idx < total / 4 && n < nc && k < p.K)
                           ? *reinterpret_cast<const float4*>(Wm + (int64_t)(n_base + n) * p.w_rs + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = base + u * 256;
                if (idx >= total / 4) continue;
                const int n = idx / (KS * 4), k = (idx % (KS * 4)) * 4;
                const uint32_t h0 = hmx_pack_bf16x2(v[u].x, v[u].y), h1 = hmx_pack_bf16x2(v[u].z, v[u].w);
                *reinterpret_cast<uint2*>(wl + n * KP + k) = make_uint2(h0, h1);
                if (SPLIT) {
                    const uint32_t l0 = hmx_pack_bf16x2(v[u].x - __uint_as_float(h0 << 16), v[u].y - __uint_as_float(h0 & 0xffff0000u));
                    const uint32_t l1 = hmx_pack_bf16x2(v[u].z - __uint_as_float(h1 << 16), v[u].w - __uint_as_float(h1 & 0xffff0000u));
                    *reinterpret_cast<uint2*>(wl_lo + n * KP + k) = make_uint2(l0, l1);
                }
            }
        }
    } else {
        constexpr int kBatch = 8;      // (32 in flight measured slower: a lane per cache line on a transposed view)
        for (int base = threadIdx.x; base < total; base += 256 * kBatch) {
            float v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = base + u * 256;
                const int n = idx / (KS * 16), k = idx % (KS * 16);
                v[u] = (idx < total && n < nc && k < p.K) ? Wm[(int64_t)(n_base + n) * p.w_rs + (int64_t)k * p.w_cs] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = base + u * 256;
                if (idx >= total) continue;
                const int n = idx / (KS * 16), k = idx % (KS * 16);
                const uint16_t hi = (uint16_t)hmx_pack_bf16x2(v[u], 0.f);
                wl[n * KP + k] = hi;
                if (SPLIT) wl_lo[n * KP + k] = (uint16_t)hmx_pack_bf16x2(v[u] - hmx_bf16_to_f32(hi), 0.f);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const float im = (ok && p.in_rowmask) ? p.in_rowmask[t] : 1.f;
        const float* xr = Xm + (ok ? t : 0) * p.ldx;
        short8 a[KS], a_lo[SPLIT ? KS : 1];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 16 * ks + 8 * h;
            float v[8];
            if (ok && k0 + 8 <= p.K && ((reinterpret_cast<uintptr_t>(xr + k0) & 15) == 0)) {
                const float4 lo = *reinterpret_cast<const float4*>(xr + k0);
                const float4 hi = *reinterpret_cast<const float4*>(xr + k0 + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
                v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (ok && k0 + e < p.K) ? xr[k0 + e] : 0.f;
            }
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // hardware pair conversion: the operand split costs ~3 ops per element
                const float x0 = v[2 * e] * im, x1 = v[2 * e + 1] * im;
                hw[e] = hmx_pack_bf16x2(x0, x1);
                if (SPLIT)
                    lw[e] = hmx_pack_bf16x2(x0 - __uint_as_float(hw[e] << 16), x1 - __uint_as_float(hw[e] & 0xffff0000u));
            }
            a[ks] = __builtin_bit_cast(short8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
            if (SPLIT) a_lo[ks] = __builtin_bit_cast(short8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
        }
        // Roles: weights are the MFMA row operand, tokens the column operand, so a lane owns ONE token (column i) and
        // sixteen outputs n = nt * 32 + 8 * q + 4 * h + (0..3), q = 0..3: four 16-byte stores per tile instead of sixteen
        // 4-byte ones, with bias and residuals read 16 bytes at a time as well.
        const int64_t tt = tile * 32 + i;
        const bool t_ok = tt < p.T;
        const float om = (t_ok && p.out_rowmask) ? p.out_rowmask[tt] : 1.f;
        const bool vec_ok = ((p.ldy | p.ldr1 | p.ldr2) & 3) == 0 &&
                            ((reinterpret_cast<uintptr_t>(Ym) | reinterpret_cast<uintptr_t>(p.R1) |
                              reinterpret_cast<uintptr_t>(p.R2) | reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0;
        for (int nt = 0; nt < npad / 32; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (nt * 32 + i) * KP + 16 * ks + 8 * h;
                const short8 b = *reinterpret_cast<const short8*>(wl + off);
                if (SPLIT) {
                    const short8 b_lo = *reinterpret_cast<const short8*>(wl_lo + off);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a_lo[ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_lo, a[ks], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a[ks], acc, 0, 0, 0);
            }
            if (!t_ok) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = n_base + nt * 32 + 8 * q + 4 * h;
                if (n0 >= n_base + nc) continue;
                float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                float* dst = Ym + tt * p.ldy + n0;
                if (vec_ok && n0 + 4 <= n_base + nc && (n0 & 3) == 0) {
                    if (p.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0);
                        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    }
                    if (p.R1) {
                        const float4 rv = *reinterpret_cast<const float4*>(p.R1 + tt * p.ldr1 + n0);
                        v[0] += p.c1 * rv.x; v[1] += p.c1 * rv.y; v[2] += p.c1 * rv.z; v[3] += p.c1 * rv.w;
                    }
                    if (p.R2) {
                        const float4 rv = *reinterpret_cast<const float4*>(p.R2 + tt * p.ldr2 + n0);
                        v[0] += p.c2 * rv.x; v[1] += p.c2 * rv.y; v[2] += p.c2 * rv.z; v[3] += p.c2 * rv.w;
                    }
                    if (p.out_rowmask) { v[0] *= om; v[1] *= om; v[2] *= om; v[3] *= om; }
                    if (p.accumulate) {
                        const float4 cur = *reinterpret_cast<const float4*>(dst);
                        v[0] += cur.x; v[1] += cur.y; v[2] += cur.z; v[3] += cur.w;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + e;
                        if (n < n_base + nc) {
                            float x = v[e] + (p.bias ? p.bias[n] : 0.f);
                            if (p.R1) x += p.c1 * p.R1[tt * p.ldr1 + n];
                            if (p.R2) x += p.c2 * p.R2[tt * p.ldr2 + n];
                            if (p.out_rowmask) x *= om;
                            if (p.accumulate) x += dst[e];
                            dst[e] = x;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- weight gradient (split-bf16 or fp32 MFMA)
// dW[n][k] (+)= sum_t dY[t][n] * X[t][k] * in_rowmask[t];  dbias[n] (+)= sum_t dY[t][n].
// One weight-gradient product of a batched launch.
struct WgradJob {
    const float* dY;
    const float* X;
    const float* in_rowmask;
    float* dW;
    float* dbias;
    int64_t ldy, ldx, ldw, T;
    int64_t slab0;           // first float of this job's slabs in the workspace
    int N, K, with_bias, accumulate;
    int tile0;               // first global macro tile of this job
    int tn, tk;              // macro tile: (32 tn) x (32 tk) outputs per wave
    int n_grp, n_grpp, k_grp, k_grpp;   // grouped outputs (0: plain), see hmx_wgrad_desc
};
constexpr int kWgradMaxJobs = 24;
struct WgradParams {
    WgradJob job[kWgradMaxJobs];
    int n_jobs;
    float* part;             // slabs [job][macro tile][group][tn * tk tile images of 16 x 64, 128 bias sums]
    unsigned* tickets;       // one arrival counter per global macro tile, zero on entry and on exit
    int fp32_mfma;           // 1: products on the fp32 MFMA; 0: operands split into two bf16 numbers, bf16 MFMA
};

// Arrival counters of the in-launch reduction below: one word per macro tile of a launch, in a slot chosen by the
// host per call.  They are zero whenever no launch is in flight: the workgroup that arrives last puts its counter
// back to zero.  (A slot belongs to one stream of one GPU at a time, see wgrad_tickets_for_call.)
constexpr int kWgradMaxTiles = 1024;
constexpr int kWgradSlots = 16;
__device__ unsigned g_wgrad_tickets[kWgradSlots * kWgradMaxTiles];

constexpr int kWgradBiasFloats = 128;      // bias partial sums at the end of a slab (32 * tn <= 128)
__host__ __device__ constexpr int64_t wgrad_slab_floats(int tn, int tk) { return (int64_t)tn * tk * 1024 + kWgradBiasFloats; }

template <int V>
__device__ __forceinline__ void wgrad_load(float (&dst)[V], const float* src) {
    if constexpr (V == 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else if constexpr (V == 2) {
        const float2 v = *reinterpret_cast<const float2*>(src);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        dst[0] = *src;
    }
}

struct WgradShared {
    float tile[4][16][64];
    float bias[4][kWgradBiasFloats];
    unsigned ticket;
};

// One macro tile of one job: a wave owns TN x TK accumulator tiles of 32 x 32 and one token split.  Lane (i, h)
// supplies, for token 2 u + h of a step, TN consecutive columns of dY and TK consecutive columns of X with ONE vector
// load each: accumulator tile (a, b) row i is output row n0 + TN i + a, column i is k0 + TK i + b.  Every operand
// element is read once per macro tile instead of once per 32 x 32 tile, which is what bounds this kernel: the operands
// of a launch do not fit L2, so each re-read comes from HBM.  The bias sum rides on the dY values in the vector unit.
template <int TN, int TK, bool BF16, int AHEAD>
__device__ __forceinline__ void wgrad_macro_tile(const WgradParams& p, const WgradJob& jb, int gtile, WgradShared& sh) {
    if constexpr (BF16 && TN * TK > 4) return;       // the host gives the bf16 form at most four tiles per wave
    constexpr int kSteps = TN * TK >= 8 ? 4 : 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int mtile = gtile - jb.tile0;
    const int mk_n = (jb.K + 32 * TK - 1) / (32 * TK);
    const int n0 = (mtile / mk_n) * 32 * TN, k0 = (mtile % mk_n) * 32 * TK;
    const int n = n0 + TN * i, k = k0 + TK * i;
    const bool n_ok = n < jb.N, k_ok = k < jb.K;       // N % TN == 0 and K % TK == 0: a lane's vector is all in or all out
    const int n_groups = gridDim.x;
    int64_t tps = hmx_ceil_div(jb.T, (int64_t)n_groups * 4);
    tps = BF16 ? (tps + 15) / 16 * 16 : tps + (tps & 1);
    const int64_t t0 = min(jb.T, ((int64_t)blockIdx.x * 4 + wave) * tps);
    const int64_t t1 = min(jb.T, t0 + tps);
    f32x16 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) bsum[a] = 0.f;
    const float* ap = jb.dY + (n_ok ? n : 0);
    const float* bp = jb.X + (k_ok ? k : 0);
    const float* mp = jb.in_rowmask;
    const int64_t ldy = jb.ldy, ldx = jb.ldx;
    if constexpr (BF16) {
        // Split-bf16 form: a trip is sixteen tokens = ONE 32 x 32 x 16 MFMA per product term.  Lane (i, h) reads, for its
        // eight tokens t + 8 h + e, the same TN / TK consecutive columns as above (one vector load per token and
        // operand: the same requests as the fp32 form), writes each column's eight values as a sum of two bf16
        // numbers (hi + lo, 16 significant bits) and the accumulator receives hi*lo + lo*hi + hi*hi: the multiply rate
        // is 5 x that of the fp32 MFMA, which leaves the operand reads as the bound.  The raw rows of the next trip
        // are in flight while this trip's fragments are multiplied.
        // AHEAD trips of raw rows (and their mask values) are in flight
        constexpr bool kMaskAhead = AHEAD > 1;       // the mask values travel with the rows where registers allow
        float ra[AHEAD][8][TN], rb[AHEAD][8][TK], rm[kMaskAhead ? AHEAD : 1][8];
        auto load16 = [&](int s, int64_t t) {       // tokens t .. t + 15, rows past t1 - 1 replaced by row t (dropped below)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t tt = t + 8 * h + e;
                const int64_t tc = tt < t1 ? tt : t;
                wgrad_load<TN>(ra[s][e], ap + tc * ldy);
                wgrad_load<TK>(rb[s][e], bp + tc * ldx);
                if constexpr (kMaskAhead) rm[s][e] = mp != nullptr ? mp[tc] : 1.f;
            }
        };
        auto split8 = [&](const float(&v)[8], short8& hi, short8& lo) {
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hw[e] = hmx_pack_bf16x2(v[2 * e], v[2 * e + 1]);
                lw[e] = hmx_pack_bf16x2(v[2 * e] - __uint_as_float(hw[e] << 16), v[2 * e + 1] - __uint_as_float(hw[e] & 0xffff0000u));
            }
            hi = __builtin_bit_cast(short8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
            lo = __builtin_bit_cast(short8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
        };
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (t0 + 16 * s < t1) load16(s, t0 + 16 * s);
        for (int64_t tb = t0; tb < t1; tb += 16 * AHEAD) {
#pragma unroll
            for (int s = 0; s < AHEAD; ++s) {
                const int64_t t = tb + 16 * s;
                if (t >= t1) break;                 // wave-uniform
                short8 ah[TN], al[TN], bh[TK], bl[TK];
                const bool whole = t + 16 <= t1;
                if constexpr (!kMaskAhead) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int64_t tt = t + 8 * h + e;
                        rm[0][e] = (mp != nullptr && tt < t1) ? mp[tt] : 1.f;
                    }
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = (n_ok && (whole || t + 8 * h + e < t1)) ? ra[s][e][a] : 0.f;
                        bsum[a] += v[e];
                    }
                    split8(v, ah[a], al[a]);
                }
#pragma unroll
                for (int b = 0; b < TK; ++b) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] = (k_ok && (whole || t + 8 * h + e < t1)) ? rb[s][e][b] * rm[kMaskAhead ? s : 0][e] : 0.f;
                    split8(v, bh[b], bl[b]);
                }
                if (t + 16 * AHEAD < t1) load16(s, t + 16 * AHEAD);
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TK; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    }
            }
        }
    } else {
        auto load_trip = [&](float(&av)[kSteps][TN], float(&bv)[kSteps][TK], int64_t t) {   // tokens t .. t + 2 kSteps - 1 < t1
    #pragma unroll
            for (int u = 0; u < kSteps; ++u) {
                const int64_t tt = t + 2 * u + h;
                wgrad_load<TN>(av[u], ap + tt * ldy);
                wgrad_load<TK>(bv[u], bp + tt * ldx);
            }
            if (mp != nullptr) {
    #pragma unroll
                for (int u = 0; u < kSteps; ++u) {
                    const float m = mp[t + 2 * u + h];
    #pragma unroll
                    for (int b = 0; b < TK; ++b) bv[u][b] *= m;
                }
            }
        };
        auto mfma_step = [&](const float(&av)[TN], const float(&bv)[TK]) {
            float az[TN], bz[TK];
    #pragma unroll
            for (int a = 0; a < TN; ++a) {
                az[a] = n_ok ? av[a] : 0.f;
                bsum[a] += az[a];
            }
    #pragma unroll
            for (int b = 0; b < TK; ++b) bz[b] = k_ok ? bv[b] : 0.f;
    #pragma unroll
            for (int a = 0; a < TN; ++a)
    #pragma unroll
                for (int b = 0; b < TK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(az[a], bz[b], acc[a][b], 0, 0, 0);
        };
        auto mfma_trip = [&](const float(&av)[kSteps][TN], const float(&bv)[kSteps][TK]) {
    #pragma unroll
            for (int u = 0; u < kSteps; ++u) mfma_step(av[u], bv[u]);
        };
        // Trips of kSteps steps through a ring of kRing operand buffers: the loads of trip j + kRing - 1 are issued before
        // the MFMAs of trip j, so two further trips are in flight while one is consumed (HBM latency under load is several
        // trips long).  The ring index is static: the loop body is kRing stages.  Loads past the last trip repeat the last
        // trip (in bounds, never consumed).
        constexpr int kRing = 3;
        const int64_t n_trips = (t1 - t0) / (2 * kSteps);
        if (n_trips > 0) {
            float av[kRing][kSteps][TN], bv[kRing][kSteps][TK];
            const int64_t last = n_trips - 1;
    #pragma unroll
            for (int s = 0; s < kRing - 1; ++s) load_trip(av[s], bv[s], t0 + min((int64_t)s, last) * (2 * kSteps));
            for (int64_t trip = 0; trip < n_trips; trip += kRing) {
    #pragma unroll
                for (int s = 0; s < kRing; ++s) {
                    load_trip(av[(s + kRing - 1) % kRing], bv[(s + kRing - 1) % kRing],
                              t0 + min(trip + s + kRing - 1, last) * (2 * kSteps));
                    if (trip + s < n_trips) mfma_trip(av[s], bv[s]);
                }
            }
        }
        int64_t t = t0 + n_trips * (2 * kSteps);
        for (; t < t1; t += 2) {                        // the split's last tokens, one step at a time
            const int64_t tt = t + h;
            const bool ok = tt < t1;
            const int64_t tc = ok ? tt : t;             // t < t1: a valid row
            float av[TN], bv[TK];
            wgrad_load<TN>(av, ap + tc * ldy);
            wgrad_load<TK>(bv, bp + tc * ldx);
            const float m = ok ? (mp ? mp[tc] : 1.f) : 0.f;
    #pragma unroll
            for (int a = 0; a < TN; ++a) av[a] = ok ? av[a] : 0.f;
    #pragma unroll
            for (int b = 0; b < TK; ++b) bv[b] *= m;
            mfma_step(av, bv);
        }
    }
    // Output of element e = r * 64 + lane_e of tile image (a, b)
    auto finish = [&](const float4& tot, int a, int b) {
        const int e0 = threadIdx.x * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q, r = e >> 6, le = e & 63;
            int nr = n0 + TN * ((r & 3) + 8 * (r >> 2) + 4 * (le >> 5)) + a, kk = k0 + TK * (le & 31) + b;
            const float val = q == 0 ? tot.x : q == 1 ? tot.y : q == 2 ? tot.z : tot.w;
            bool ok = nr < jb.N && kk < jb.K;
            if (jb.n_grpp > 0) {
                const int g = nr / jb.n_grpp, w = nr % jb.n_grpp;
                ok = ok && w < jb.n_grp;
                nr = g * jb.n_grp + w;
            }
            if (jb.k_grpp > 0) {
                const int g = kk / jb.k_grpp, w = kk % jb.k_grpp;
                ok = ok && w < jb.k_grp;
                kk = g * jb.k_grp + w;
            }
            if (ok) {
                float* dst = jb.dW + (int64_t)nr * jb.ldw + kk;
                *dst = jb.accumulate ? *dst + val : val;
            }
        }
    };
    auto finish_bias = [&](float val) {             // thread a * 32 + i owns output row n0 + TN i + a
        const int a = threadIdx.x >> 5, ii = threadIdx.x & 31;
        int nr = n0 + TN * ii + a;
        bool ok = nr < jb.N;
        if (jb.n_grpp > 0) {
            const int g = nr / jb.n_grpp, w = nr % jb.n_grpp;
            ok = ok && w < jb.n_grp;
            nr = g * jb.n_grp + w;
        }
        if (ok) jb.dbias[nr] = jb.accumulate ? jb.dbias[nr] + val : val;
    };
    const bool bias_here = jb.with_bias && k0 == 0;
    const int64_t slab_f = wgrad_slab_floats(TN, TK);
    float* slabs = p.part + jb.slab0 + (int64_t)mtile * n_groups * slab_f;     // [group][slab] of this macro tile
    float* my_slab = slabs + (int64_t)blockIdx.x * slab_f;
    // the four waves' tiles, added in wave order, one tile image at a time
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) {
            if (a + b > 0) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) sh.tile[wave][r][lane] = acc[a][b][r];
            __syncthreads();
            const float* s0 = &sh.tile[0][0][0] + threadIdx.x * 4;
            float4 v = *reinterpret_cast<const float4*>(s0);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = *reinterpret_cast<const float4*>(s0 + w * 1024);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (n_groups == 1) finish(v, a, b);
            else reinterpret_cast<float4*>(my_slab)[(a * TK + b) * 256 + threadIdx.x] = v;
        }
    if (bias_here) {
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const float s = bsum[a] + __shfl_xor(bsum[a], 32);
            if (h == 0) sh.bias[wave][a * 32 + i] = s;
        }
        __syncthreads();
        if (threadIdx.x < 32 * TN) {
            const float s = ((sh.bias[0][threadIdx.x] + sh.bias[1][threadIdx.x]) + sh.bias[2][threadIdx.x]) + sh.bias[3][threadIdx.x];
            if (n_groups == 1) finish_bias(s);
            else my_slab[TN * TK * 1024 + threadIdx.x] = s;
        }
    }
    if (n_groups == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.ticket = __hip_atomic_fetch_add(p.tickets + gtile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (sh.ticket != (unsigned)(n_groups - 1)) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(p.tickets + gtile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // last to arrive: the slabs of every group, added in group order; eight groups of up to four of the macro tile's
    // images are requested together (32 reads in flight per lane: the chain of round trips is the tail of the launch)
    constexpr int kImgs = TN * TK < 4 ? TN * TK : 4;
#pragma unroll
    for (int ab0 = 0; ab0 < TN * TK; ab0 += kImgs) {
        const float4* src = reinterpret_cast<const float4*>(slabs) + ab0 * 256 + threadIdx.x;
        const int64_t stride = slab_f / 4;
        float4 tot[kImgs];
#pragma unroll
        for (int ab = 0; ab < kImgs; ++ab) tot[ab] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < n_groups; g += 8) {
            float4 o[kImgs][8];
#pragma unroll
            for (int ab = 0; ab < kImgs; ++ab)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    o[ab][u] = g + u < n_groups ? src[(int64_t)(g + u) * stride + ab * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ab = 0; ab < kImgs; ++ab)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    tot[ab].x += o[ab][u].x; tot[ab].y += o[ab][u].y; tot[ab].z += o[ab][u].z; tot[ab].w += o[ab][u].w;
                }
        }
#pragma unroll
        for (int ab = 0; ab < kImgs; ++ab) finish(tot[ab], (ab0 + ab) / TK, (ab0 + ab) % TK);
    }
    if (bias_here && threadIdx.x < 32 * TN) {
        float tot = 0.f;
        for (int g = 0; g < n_groups; ++g) tot += slabs[(int64_t)g * slab_f + TN * TK * 1024 + threadIdx.x];
        finish_bias(tot);
    }
}

template <bool BF16, bool SMALL>
__device__ __forceinline__ void wgrad_dispatch(const WgradParams& p, const WgradJob& jb, int gtile, WgradShared& sh) {
    switch (jb.tn * 8 + jb.tk) {
        case 4 * 8 + 2: if constexpr (!SMALL) wgrad_macro_tile<4, 2, BF16, (SMALL || 4 * 2 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 2 * 8 + 4: if constexpr (!SMALL) wgrad_macro_tile<2, 4, BF16, (SMALL || 2 * 4 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 4 * 8 + 1: wgrad_macro_tile<4, 1, BF16, (SMALL || 4 * 1 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 1 * 8 + 4: wgrad_macro_tile<1, 4, BF16, (SMALL || 1 * 4 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 2 * 8 + 2: wgrad_macro_tile<2, 2, BF16, (SMALL || 2 * 2 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 2 * 8 + 1: wgrad_macro_tile<2, 1, BF16, (SMALL || 2 * 1 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        case 1 * 8 + 2: wgrad_macro_tile<1, 2, BF16, (SMALL || 1 * 2 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
        default: wgrad_macro_tile<1, 1, BF16, (SMALL || 1 * 1 > 4) ? 1 : 2>(p, jb, gtile, sh); break;
    }
}

// Weight gradients of up to kWgradMaxJobs projections in one launch.  blockIdx.y walks the macro tiles of all jobs,
// blockIdx.x the token-split groups.  One workgroup = four waves = four consecutive token splits of ONE macro tile.  A
// wave runs trips of MFMA steps with the loads of the next trip issued before the MFMAs of this one.
// The four partial results are added in LDS (wave order) and written as one slab; the workgroup that arrives last at
// the tile's counter adds the slabs in group order and writes (or adds to) dW / dbias: fixed association order, no
// float atomics, no second launch.
// SMALL: every job of the launch has at most four accumulator tiles per wave; the kernel is then built for three
// workgroups per CU instead of two (the operand reads in flight per CU are what bounds it).
template <bool SMALL>
__global__ __launch_bounds__(256, SMALL ? 3 : 2) void wgrad_kernel(WgradParams p) {
    __shared__ WgradShared sh;
    const int gtile = blockIdx.y;
    int ji = 0;
    while (ji + 1 < p.n_jobs && p.job[ji + 1].tile0 <= gtile) ++ji;     // wave-uniform scan of at most 24 entries
    const WgradJob jb = p.job[ji];      // a copy in scalar registers (through a reference the argument block went to scratch)
    if constexpr (SMALL) {
        wgrad_dispatch<true, true>(p, jb, gtile, sh);          // the host launches this form for the bf16 MFMA only
    } else {
        if (p.fp32_mfma) wgrad_dispatch<false, false>(p, jb, gtile, sh);
        else wgrad_dispatch<true, false>(p, jb, gtile, sh);
    }
}

// ---------------------------------------------------------------------------- pointwise kernels
// s = SiLU(x1);  y2 = RoPE(s) on interleaved pairs, angle tables (L, hd/2)   (hydra3.py:368-386)
__global__ __launch_bounds__(256) void silu_rope_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ cs,
                                                            const float* __restrict__ sn, int64_t T, int L, int d,
                                                            int hd, float* __restrict__ s, float* __restrict__ y2) {
    const int64_t total = T * (d / 2);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / (d / 2);
        const int pp = (int)(idx % (d / 2)), ip = pp % (hd / 2), pos = (int)(t % L);
        const float c = cs[pos * (hd / 2) + ip], sv = sn[pos * (hd / 2) + ip];
        const float2 xv = *reinterpret_cast<const float2*>(x1 + t * d + 2 * pp);
        const float se = silu_(xv.x), so = silu_(xv.y);
        *reinterpret_cast<float2*>(s + t * d + 2 * pp) = make_float2(se, so);
        *reinterpret_cast<float2*>(y2 + t * d + 2 * pp) = make_float2(se * c - so * sv, se * sv + so * c);
    }
}

// dx1 = (ds + RoPE^T(dy2)) * SiLU'(x1)
__global__ __launch_bounds__(256) void silu_rope_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ cs,
                                                            const float* __restrict__ sn, const float* __restrict__ ds,
                                                            const float* __restrict__ dy2, int64_t T, int L, int d,
                                                            int hd, float* __restrict__ dx1) {
    const int64_t total = T * (d / 2);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / (d / 2);
        const int pp = (int)(idx % (d / 2)), ip = pp % (hd / 2), pos = (int)(t % L);
        const float c = cs[pos * (hd / 2) + ip], sv = sn[pos * (hd / 2) + ip];
        const float2 xv = *reinterpret_cast<const float2*>(x1 + t * d + 2 * pp);
        const float2 g1 = *reinterpret_cast<const float2*>(ds + t * d + 2 * pp);
        const float2 g2 = *reinterpret_cast<const float2*>(dy2 + t * d + 2 * pp);
        const float ge = g1.x + g2.x * c + g2.y * sv;
        const float go = g1.y - g2.x * sv + g2.y * c;
        *reinterpret_cast<float2*>(dx1 + t * d + 2 * pp) = make_float2(ge * silu_grad_(xv.x), go * silu_grad_(xv.y));
    }
}

// mode 0: out = a * b * c                      (y1 * y2 / sqrt(H),           hydra3.py:387)
// mode 1: out = SiLU(a) * b                    (SwiGLU gate,                 hydra3.py:416-418)
__global__ __launch_bounds__(256) void binary_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t n, float c, int mode, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = mode == 0 ? a[i] * b[i] * c : silu_(a[i]) * b[i];
}

__global__ __launch_bounds__(256) void binary_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ g, int64_t n, float c, int mode,
                                                         float* __restrict__ da, float* __restrict__ db) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mode == 0) {
            da[i] = g[i] * b[i] * c;
            db[i] = g[i] * a[i] * c;
        } else {
            da[i] = g[i] * b[i] * silu_grad_(a[i]);
            db[i] = g[i] * silu_(a[i]);
        }
    }
}

// 16-byte forms of the three kernels above / below (n % 4 == 0, 16-byte aligned pointers; d % 4 == 0 for combine).
__global__ __launch_bounds__(256) void binary_fwd_vec4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                              int64_t n4, float c, int mode, float4* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        out[i] = mode == 0 ? make_float4(x.x * y.x * c, x.y * y.y * c, x.z * y.z * c, x.w * y.w * c)
                           : make_float4(silu_(x.x) * y.x, silu_(x.y) * y.y, silu_(x.z) * y.z, silu_(x.w) * y.w);
    }
}
__global__ __launch_bounds__(256) void binary_bwd_vec4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                              const float4* __restrict__ g, int64_t n4, float c, int mode,
                                                              float4* __restrict__ da, float4* __restrict__ db) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], gg = g[i];
        if (mode == 0) {
            da[i] = make_float4(gg.x * y.x * c, gg.y * y.y * c, gg.z * y.z * c, gg.w * y.w * c);
            db[i] = make_float4(gg.x * x.x * c, gg.y * x.y * c, gg.z * x.z * c, gg.w * x.w * c);
        } else {
            da[i] = make_float4(gg.x * y.x * silu_grad_(x.x), gg.y * y.y * silu_grad_(x.y), gg.z * y.z * silu_grad_(x.z),
                                gg.w * y.w * silu_grad_(x.w));
            db[i] = make_float4(gg.x * silu_(x.x), gg.y * silu_(x.y), gg.z * silu_(x.z), gg.w * silu_(x.w));
        }
    }
}
__global__ __launch_bounds__(256) void combine_vec4_kernel(const float4* __restrict__ x0, const float4* __restrict__ x1,
                                                           const float4* __restrict__ x2, const float4* __restrict__ x3,
                                                           float c0, float c1, float c2, float c3,
                                                           const float* __restrict__ rowmask, int64_t T, int d4,
                                                           float4* __restrict__ out) {
    const int64_t n = T * d4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float4 a = x0[i];
        float4 v = make_float4(c0 * a.x, c0 * a.y, c0 * a.z, c0 * a.w);   // same operation order per element as the scalar form
        if (x1) { const float4 t = x1[i]; v.x += c1 * t.x; v.y += c1 * t.y; v.z += c1 * t.z; v.w += c1 * t.w; }
        if (x2) { const float4 t = x2[i]; v.x += c2 * t.x; v.y += c2 * t.y; v.z += c2 * t.z; v.w += c2 * t.w; }
        if (x3) { const float4 t = x3[i]; v.x += c3 * t.x; v.y += c3 * t.y; v.z += c3 * t.z; v.w += c3 * t.w; }
        if (rowmask) { const float m = rowmask[i / d4]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
        out[i] = v;
    }
}

// out[t, :] = (c0 X0 + c1 X1 + c2 X2 + c3 X3)[t, :] * rowmask[t]   (null inputs are skipped)
__global__ __launch_bounds__(256) void combine_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                      const float* __restrict__ x2, const float* __restrict__ x3,
                                                      float c0, float c1, float c2, float c3,
                                                      const float* __restrict__ rowmask, int64_t T, int d,
                                                      float* __restrict__ out) {
    const int64_t n = T * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = c0 * x0[i];
        if (x1) v += c1 * x1[i];
        if (x2) v += c2 * x2[i];
        if (x3) v += c3 * x3[i];
        if (rowmask) v *= rowmask[i / d];
        out[i] = v;
    }
}

// Index preparation of log2feats (hydra3.py:621-659): padding mask, position ids, time-interval bins.
__global__ __launch_bounds__(256) void seq_index_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ ts,
                                                        int64_t T, int L, int time_span, float* __restrict__ mask,
                                                        int64_t* __restrict__ pos_idx, int64_t* __restrict__ time_bin) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (int64_t)gridDim.x * 256) {
        const int l = (int)(t % L);
        const bool valid = ids[t] != 0;
        mask[t] = valid ? 1.f : 0.f;
        pos_idx[t] = valid ? l + 1 : 0;
        int64_t dt = 0;
        if (l > 0 && ids[t - 1] != 0) dt = ts[t] - ts[t - 1];
        int64_t bin = (int64_t)logf((float)dt + 1.0f);
        bin = bin > time_span - 1 ? time_span - 1 : bin;
        time_bin[t] = valid ? bin : 0;
    }
}

// ---------------------------------------------------------------------------- causal depthwise conv + SiLU
// x: (T, C).  xc[t, c] = SiLU(b[c] + sum_j w[c][j] x[t-3+j, c]) with taps before the sequence start reading
// zero   (hydra3.py:243-250, 266-269).
__global__ __launch_bounds__(256) void conv_silu_fwd_kernel(const float* __restrict__ xz, const float* __restrict__ w,
                                                            const float* __restrict__ b, int64_t T, int L, int C,
                                                            float* __restrict__ xc) {
    const int64_t n = T * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / C;
        const int c = (int)(idx % C), pos = (int)(t % L);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tp = pos - 3 + j;
            if (tp >= 0) acc += w[c * 4 + j] * xz[(t - 3 + j) * C + c];
        }
        xc[idx] = silu_(b[c] + acc);
    }
}

// Same, four channels and kConvRun consecutive positions of one sequence per thread (16-byte loads and stores, every
// input row read by one thread apart from the three before its run); C % 4 == 0 and 16-byte aligned rows.
constexpr int kConvRun = 8;
__global__ __launch_bounds__(256) void conv_silu_fwd_vec4_kernel(const float* __restrict__ xz, const float* __restrict__ w,
                                                                 const float* __restrict__ b, int64_t T, int L, int C,
                                                                 float* __restrict__ xc) {
    const int C4 = C / 4;
    const int runs = (L + kConvRun - 1) / kConvRun;
    const int64_t n = T / L * runs * C4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int c = 4 * (int)(idx % C4);
        const int64_t sr = idx / C4;
        const int p0 = (int)(sr % runs) * kConvRun;          // first position of the run
        const int64_t row0 = sr / runs * L + p0;
        float4 wv[4];   // wv[i] = taps of channel c + i
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = *reinterpret_cast<const float4*>(w + (c + i) * 4);
        const float4 bv = *reinterpret_cast<const float4*>(b + c);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 x[kConvRun + 3];   // x[k] = input at position p0 - 3 + k; zero before the sequence start
#pragma unroll
        for (int k = 0; k < kConvRun + 3; ++k) {
            const int pos = p0 - 3 + k;
            x[k] = (pos >= 0 && pos < L) ? *reinterpret_cast<const float4*>(xz + (row0 - 3 + k) * C + c) : zero;
        }
#pragma unroll
        for (int k = 0; k < kConvRun; ++k) {
            if (p0 + k >= L) break;
            // per channel: taps in ascending order from zero, as the scalar kernel adds them (a tap before the
            // sequence start, skipped there, adds a zero product here: the same sum)
            float4 acc = zero;
            acc.x += wv[0].x * x[k].x; acc.y += wv[1].x * x[k].y; acc.z += wv[2].x * x[k].z; acc.w += wv[3].x * x[k].w;
            acc.x += wv[0].y * x[k + 1].x; acc.y += wv[1].y * x[k + 1].y; acc.z += wv[2].y * x[k + 1].z; acc.w += wv[3].y * x[k + 1].w;
            acc.x += wv[0].z * x[k + 2].x; acc.y += wv[1].z * x[k + 2].y; acc.z += wv[2].z * x[k + 2].z; acc.w += wv[3].z * x[k + 2].w;
            acc.x += wv[0].w * x[k + 3].x; acc.y += wv[1].w * x[k + 3].y; acc.z += wv[2].w * x[k + 3].z; acc.w += wv[3].w * x[k + 3].w;
            float4 o;
            o.x = silu_(bv.x + acc.x);
            o.y = silu_(bv.y + acc.y);
            o.z = silu_(bv.z + acc.z);
            o.w = silu_(bv.w + acc.w);
            *reinterpret_cast<float4*>(xc + (row0 + k) * C + c) = o;
        }
    }
}

// One block per (sequence, 256-channel slab); walks the sequence backwards with sliding windows.
// Writes dx (T, C) and per-sequence partials part[b] = [dw (C, 4) | db (C)].
__global__ __launch_bounds__(256) void conv_silu_bwd_kernel(const float* __restrict__ xz, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ dxc,
                                                            int L, int C, float* __restrict__ dxz,
                                                            float* __restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t seq = blockIdx.x;
    const float w0 = w[c * 4], w1 = w[c * 4 + 1], w2 = w[c * 4 + 2], w3 = w[c * 4 + 3], bb = b[c];
    const float* xs = xz + seq * L * C + c;
    auto X = [&](int t) { return t >= 0 ? xs[(int64_t)t * C] : 0.f; };
    float x0 = X(L - 4), x1 = X(L - 3), x2 = X(L - 2), x3 = X(L - 1);  // x[t-3..t]
    float du1 = 0.f, du2 = 0.f, du3 = 0.f;                             // du[t+1..t+3]
    float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f, dw3 = 0.f, db = 0.f;
    auto one_step = [&](int t, float g, float x_new) {
        const float u = bb + (((w0 * x0) + w1 * x1) + w2 * x2) + w3 * x3;
        const float du = g * silu_grad_(u);
        dw0 += du * x0; dw1 += du * x1; dw2 += du * x2; dw3 += du * x3;
        db += du;
        dxz[(seq * L + t) * C + c] = w3 * du + w2 * du1 + w1 * du2 + w0 * du3;
        du3 = du2; du2 = du1; du1 = du;
        x3 = x2; x2 = x1; x1 = x0; x0 = x_new;
    };
    int t = L - 1;
    for (; t >= 3; t -= 4) {   // four steps per trip: their eight loads are requested together, the steps stay in order
        float g[4], xn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = dxc[(seq * L + t - k) * C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) xn[k] = X(t - 4 - k);
#pragma unroll
        for (int k = 0; k < 4; ++k) one_step(t - k, g[k], xn[k]);
    }
    for (; t >= 0; --t) one_step(t, dxc[(seq * L + t) * C + c], X(t - 4));
    float* o = part + seq * C * 5;
    *reinterpret_cast<float4*>(o + c * 4) = make_float4(dw0, dw1, dw2, dw3);
    o[4 * C + c] = db;
}

// ---------------------------------------------------------------------------- selective scan
// Per channel c (head hh = c / ED):  delta = softplus(b_dt + W_dt . dr_t),  a = exp(delta * A),
//   h = a * h + (delta * B_t) * x_t,  y_t = sum_n h[n] C_t[n] + D x_t,  out = y_t * SiLU(z_t)
// (hydra3.py:107-123, 203-214, 273).  dbc rows (T, H, 48) = [dr(16) | B(16) | C(16)] are staged in LDS.
struct ScanParams {
    const float* xc;     // (T, C)
    const float* xz;     // z gate input, (T, C)
    const float* dbc;    // (T, H, 48)
    const float* Wdt;    // (C, 16)
    const float* bdt;    // (C)
    const float* A_log;  // (C, 16)
    const float* D;      // (C)
    int B, L, C, ED, H;
    float* yg;           // (T, C)
    float* hs;           // (B, n_ck, 4, C, 4) or null: state checkpoints for the backward pass, one every kCk steps:
                         // checkpoint j of a sequence is the state after step kCk * (j + 1) - 1 (the state that enters
                         // sub-chunk j + 1), n_ck = (L - 1) / kCk; state n of channel c at [n / 4][c][n % 4], so that
                         // one 16-byte store per lane writes whole cache lines per wave.  The backward pass recomputes
                         // the states in between (kCk steps at a time, parked in LDS) instead of reading them back:
                         // 1 / kCk of the state traffic in both directions.
};

constexpr float kLog2e = 1.44269504088896341f, kLn2 = 0.693147180559945309f;
constexpr int kScanChunk = 64;  // time steps of dbc staged per LDS fill (forward)
constexpr int kCk = 4;          // steps between state checkpoints
constexpr int kScanChunkBwd = 32;  // backward: dbc rows per LDS fill (a multiple of kCk; leaves room for the state history)
__host__ __device__ inline int scan_n_ck(int L) { return L > 1 ? (L - 1) / kCk : 0; }

__global__ __launch_bounds__(256) void scan_fwd_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rows = reinterpret_cast<float*>(smem);  // [kScanChunk][H][48]
    const int c = blockIdx.y * 256 + threadIdx.x;
    const bool live = c < p.C;
    const int cc = live ? c : 0;
    const int hh = cc / p.ED;
    const int64_t seq = blockIdx.x;
    const int HP = p.H * kP;
    float A2[kN], Wd[kR], h[kN];   // A2 = A * log2(e): the decay exp(delta A) is one multiply and one v_exp_f32
#pragma unroll
    for (int n = 0; n < kN; ++n) {
        A2[n] = -__expf(p.A_log[cc * kN + n]) * kLog2e;
        Wd[n] = p.Wdt[cc * kR + n];
        h[n] = 0.f;
    }
    const float bd = p.bdt[cc], Dv = p.D[cc];
    for (int tc0 = 0; tc0 < p.L; tc0 += kScanChunk) {
        const int steps = min(kScanChunk, p.L - tc0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < steps * HP; idx += 256) rows[idx] = p.dbc[(seq * p.L + tc0) * HP + idx];
        __syncthreads();
        if (!live) continue;
        // x and z of the next step are requested while this step computes (unconditional, clamped at the chunk end)
        float nx = p.xc[(seq * p.L + tc0) * p.C + c];
        float nz = p.xz ? p.xz[(seq * p.L + tc0) * p.C + c] : 0.f;
        for (int s = 0; s < steps; ++s) {
            const int64_t row = seq * p.L + tc0 + s;
            const float x = nx, zv = nz;
            {
                const int64_t rn = seq * p.L + tc0 + min(s + 1, steps - 1);
                nx = p.xc[rn * p.C + c];
                nz = p.xz ? p.xz[rn * p.C + c] : 0.f;
            }
            const float* r = rows + s * HP + hh * kP;
            float dpre = 0.f;
#pragma unroll
            for (int k = 0; k < kR; ++k) dpre += Wd[k] * r[k];
            const float delta = softplus_(dpre + bd);
            const float dxv = delta * x;
            float y = 0.f;
#pragma unroll
            for (int n = 0; n < kN; ++n) {
                const float a = __builtin_amdgcn_exp2f(delta * A2[n]);
                h[n] = a * h[n] + dxv * r[kR + n];
                y += h[n] * r[kR + kN + n];
            }
            const int tl = tc0 + s;   // position in the sequence
            if (p.hs && ((tl + 1) % kCk) == 0 && tl + 1 < p.L) {
                float4* dst = reinterpret_cast<float4*>(p.hs) + (seq * scan_n_ck(p.L) + (tl + 1) / kCk - 1) * 4 * p.C + c;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[(int64_t)q * p.C] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
            }
            p.yg[row * p.C + c] = (y + Dv * x) * (p.xz ? silu_(zv) : 1.f);
        }
    }
}

// Sum of v[n] over the G lanes of a channel group, for all 16 n at once, by a halving butterfly:
// afterwards lane l of the group (with l % (G/16) == 0) holds the total for n = l / (G/16).
// x + (x of the partner lane) where the lower lane of a pair keeps x and the upper keeps y, for pairs 32 or 16 lanes
// apart: the gfx950 lane-swap instructions exchange the two halves between two registers in one VALU operation
// (no LDS crossbar, no selects), after which both lanes hold their two addends.
template <int OFF>
__device__ __forceinline__ float pair_swap_add(float x, float y) {
    static_assert(OFF == 32 || OFF == 16, "lane-swap distances");
    // Written as inline assembly with both registers read-write: this compiler's builtin for the instruction hands
    // back the first result register for both outputs.  The leading s_nop covers the wait states the instruction
    // needs after a VALU write of its operands (the compiler inserts the same before its own uses).
    if constexpr (OFF == 32)
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    else
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
template <int OFF>
__device__ __forceinline__ float pair_add(float lower_keeps, float upper_keeps) {
    if constexpr (OFF == 32 || OFF == 16) {
        return pair_swap_add<OFF>(lower_keeps, upper_keeps);
    } else {
        const bool up = (threadIdx.x & OFF) != 0;
        return (up ? upper_keeps : lower_keeps) + __shfl_xor(up ? lower_keeps : upper_keeps, OFF);
    }
}

template <int G>
__device__ __forceinline__ float group_reduce16(const float (&v)[16]) {
    float w8[8], w4[4], w2[2], w1;
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = pair_add<G / 2>(v[i], v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = pair_add<G / 4>(w8[i], w8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = pair_add<G / 8>(w4[i], w4[i + 2]);
    w1 = pair_add<G / 16>(w2[0], w2[1]);
#pragma unroll
    for (int off = G / 32; off >= 1; off >>= 1) w1 += __shfl_xor(w1, off);
    return w1;
}

struct ScanBwdParams {
    ScanParams f;          // forward inputs (yg unused); f.hs = saved states
    const float* dyg;      // (T, C)
    float* dxc;            // (T, C)
    float* dxz;            // dz, (T, C)
    float* ddbc_part;      // (T, H, GPH, 48): per channel-group partial sums of d(dbc)
    float* par_part;       // (B, C, 34): [dA_log(16) | dWdt(16) | dbdt | dD] per sequence
    int GPH;               // channel groups per head = ED / G
};

template <int G>
__global__ __launch_bounds__(256, 2) void scan_bwd_kernel(ScanBwdParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [kCk][4][256] float4: the states of the current sub-chunk, recomputed from its checkpoint; then the dbc rows
    float4* hist = reinterpret_cast<float4*>(smem);
    float* rows = reinterpret_cast<float*>(smem + (size_t)kCk * 4 * 256 * sizeof(float4));
    const ScanParams& p = q.f;
    const int c = blockIdx.y * 256 + threadIdx.x;   // C is a multiple of G (host-checked), so whole groups are live
    const bool live = c < p.C;
    const int cc = live ? c : 0;
    const int hh = cc / p.ED, ce = cc % p.ED;
    const int gih = ce / G, lig = ce % G;
    const int64_t seq = blockIdx.x;
    const int HP = p.H * kP;
    const int n_ck = scan_n_ck(p.L);
    float A2[kN], Wd[kR], carry[kN], dA[kN], dWd[kR];   // A2 = A * log2(e) as in the forward kernel
#pragma unroll
    for (int n = 0; n < kN; ++n) {
        A2[n] = -__expf(p.A_log[cc * kN + n]) * kLog2e;
        Wd[n] = p.Wdt[cc * kR + n];
        carry[n] = 0.f;
        dA[n] = 0.f;
        dWd[n] = 0.f;
    }
    const float bd = p.bdt[cc], Dv = p.D[cc];
    float dbd = 0.f, dD = 0.f;
    const int n_chunks = (p.L + kScanChunkBwd - 1) / kScanChunkBwd;
    const float4* ckp = reinterpret_cast<const float4*>(p.hs);
    // Operands of one sub-chunk (kCk steps): x, z, dy of its steps and the checkpoint that enters it.  They are
    // requested one sub-chunk ahead (unconditionally, clamped at the sequence start) so that their latency hides
    // behind the recompute + backward arithmetic of the current sub-chunk.
    float nx[kCk], nz[kCk], ngy[kCk];
    float4 nh[4];
    auto request = [&](int t0) {     // sub-chunk starting at position t0 (a multiple of kCk), t0 >= 0
#pragma unroll
        for (int u = 0; u < kCk; ++u) {
            const int64_t rr = seq * p.L + min(t0 + u, p.L - 1);
            nx[u] = p.xc[rr * p.C + c];
            nz[u] = p.xz ? p.xz[rr * p.C + c] : 0.f;
            ngy[u] = q.dyg[rr * p.C + c];
        }
        const int j = t0 / kCk - 1;   // checkpoint entering the sub-chunk (none for the first: zeros)
        const float4* src = ckp + (seq * n_ck + max(j, 0)) * 4 * p.C + c;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) nh[k4] = (n_ck > 0) ? src[(int64_t)k4 * p.C] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int t_last = ((p.L - 1) / kCk) * kCk;     // start of the last sub-chunk
    if (live) request(t_last);
    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int tc0 = ch * kScanChunkBwd;
        const int steps = min(kScanChunkBwd, p.L - tc0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < steps * HP; idx += 256) rows[idx] = p.dbc[(seq * p.L + tc0) * HP + idx];
        __syncthreads();
        if (!live) continue;
        for (int s0 = ((steps - 1) / kCk) * kCk; s0 >= 0; s0 -= kCk) {   // sub-chunks of this fill, last first
            const int t0 = tc0 + s0;
            const int ns = min(kCk, p.L - t0);
            float x[kCk], z[kCk], gy[kCk], h[kN];
#pragma unroll
            for (int u = 0; u < kCk; ++u) {
                x[u] = nx[u];
                z[u] = nz[u];
                gy[u] = ngy[u];
            }
            const bool first = t0 == 0;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                h[4 * k4] = first ? 0.f : nh[k4].x; h[4 * k4 + 1] = first ? 0.f : nh[k4].y;
                h[4 * k4 + 2] = first ? 0.f : nh[k4].z; h[4 * k4 + 3] = first ? 0.f : nh[k4].w;
            }
            request(max(t0 - kCk, 0));
            // forward recompute of the sub-chunk's states (same operations, same order as scan_fwd_kernel)
            float dpre_s[kCk];
#pragma unroll
            for (int u = 0; u < kCk; ++u) {
                if (u < ns) {
                    const float* r = rows + (s0 + u) * HP + hh * kP;
                    float dpre = 0.f;
#pragma unroll
                    for (int k = 0; k < kR; ++k) dpre += Wd[k] * r[k];
                    const float delta = softplus_(dpre + bd);
                    dpre_s[u] = dpre + bd;
                    const float dxv = delta * x[u];
#pragma unroll
                    for (int n = 0; n < kN; ++n) {
                        const float a = __builtin_amdgcn_exp2f(delta * A2[n]);
                        h[n] = a * h[n] + dxv * r[kR + n];
                    }
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        hist[(u * 4 + k4) * 256 + threadIdx.x] = make_float4(h[4 * k4], h[4 * k4 + 1], h[4 * k4 + 2], h[4 * k4 + 3]);
                }
            }
            // backward over the sub-chunk (a thread reads back only what it wrote itself: no barrier needed)
#pragma unroll
            for (int u = kCk - 1; u >= 0; --u) {
                if (u >= ns) continue;
                const int64_t row = seq * p.L + t0 + u;
                const float* r = rows + (s0 + u) * HP + hh * kP;
                float hcur[kN];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 v = hist[(u * 4 + k4) * 256 + threadIdx.x];
                    hcur[4 * k4] = v.x; hcur[4 * k4 + 1] = v.y; hcur[4 * k4 + 2] = v.z; hcur[4 * k4 + 3] = v.w;
                }
                const float dpre = dpre_s[u];
                const float delta = softplus_(dpre);
                float y = 0.f;
#pragma unroll
                for (int n = 0; n < kN; ++n) y += hcur[n] * r[kR + kN + n];
                const float ypre = y + Dv * x[u];
                const float dy = p.xz ? gy[u] * silu_(z[u]) : gy[u];
                if (p.xz) q.dxz[row * p.C + c] = gy[u] * ypre * silu_grad_(z[u]);
                dD += dy * x[u];
                // With gh = g (a h_{t-1}):  dA += delta gh,  d delta = sum_n (A gh + g B x) = ln2 sum A2 gh + x sum g B,
                // dx = D dy + delta sum g B:  the two sums are shared, A itself is not needed in the loop.
                const float dxv = delta * x[u];
                float sgh = 0.f, sgb = 0.f;
                float vB[kN], vC[kN];
#pragma unroll
                for (int n = 0; n < kN; ++n) {
                    const float a = __builtin_amdgcn_exp2f(delta * A2[n]);
                    const float Bn = r[kR + n];
                    const float g = r[kR + kN + n] * dy + carry[n];   // dL/dh_t[n]
                    carry[n] = a * g;
                    const float gh = g * (hcur[n] - dxv * Bn);        // g * a_t * h_{t-1}
                    dA[n] += delta * gh;
                    sgh += A2[n] * gh;
                    sgb += g * Bn;
                    vB[n] = g * dxv;
                    vC[n] = dy * hcur[n];
                }
                q.dxc[row * p.C + c] = Dv * dy + delta * sgb;
                const float ddelta = kLn2 * sgh + x[u] * sgb;
                const float dd = ddelta * (dpre > 20.f ? 1.f : sigmoid_(dpre));
                dbd += dd;
                float vR[kR];
#pragma unroll
                for (int k = 0; k < kR; ++k) {
                    dWd[k] += dd * r[k];
                    vR[k] = dd * Wd[k];
                }
                const float sR = group_reduce16<G>(vR);
                const float sB = group_reduce16<G>(vB);
                const float sC = group_reduce16<G>(vC);
                if ((lig & (G / 16 - 1)) == 0) {
                    const int n = lig / (G / 16);
                    float* o = q.ddbc_part + ((row * p.H + hh) * q.GPH + gih) * kP;
                    o[n] = sR;
                    o[kR + n] = sB;
                    o[kR + kN + n] = sC;
                }
            }
        }
    }
    if (live) {
        // one block per parameter: [dA_log (C, 16) | dW_dt (C, 16) | db_dt (C) | dD (C)]
        float* o = q.par_part + seq * p.C * 34;
        float* oa = o + c * kN;
        float* ow = o + (int64_t)p.C * kN + c * kR;
#pragma unroll
        for (int n = 0; n < kN; n += 4) {   // A = -exp(A_log)  =>  dA/dA_log = A = A2 ln2
            *reinterpret_cast<float4*>(oa + n) = make_float4(dA[n] * (A2[n] * kLn2), dA[n + 1] * (A2[n + 1] * kLn2),
                                                             dA[n + 2] * (A2[n + 2] * kLn2), dA[n + 3] * (A2[n + 3] * kLn2));
            *reinterpret_cast<float4*>(ow + n) = make_float4(dWd[n], dWd[n + 1], dWd[n + 2], dWd[n + 3]);
        }
        o[(int64_t)p.C * 32 + c] = dbd;
        o[(int64_t)p.C * 33 + c] = dD;
    }
}

// out[r][j] = sum_g part[r][g][j]  (fixed order)
__global__ __launch_bounds__(256) void sum_groups_kernel(const float* __restrict__ part, int64_t rows, int groups, int w,
                                                         float* __restrict__ out) {
    const int64_t n = rows * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / w;
        const int j = (int)(i % w);
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += part[(r * groups + g) * w + j];
        out[i] = s;
    }
}

int ks_for(int K) {
    int ks = 1;
    while (ks * 16 < K) ks *= 2;
    return ks;
}

}  // namespace

// ================================================================================ C ABI
extern "C" int hmx_rmsnorm_fwd(const float* x, const float* scale, int64_t T, int64_t d, float eps, float* y,
                               float* rstd, void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x || !scale || !y || d <= 0 || d > 256) return HMX_EINVAL;
    if (d % 4 == 0 && aligned16(x, scale, y))
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<4>, dim3(grid_for(T, 16)), dim3(256), 0, (hipStream_t)stream, x, scale, T,
                           (int)d, eps, y, rstd);
    else
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<1>, dim3(grid_for(T, 16)), dim3(256), 0, (hipStream_t)stream, x, scale, T,
                           (int)d, eps, y, rstd);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int64_t hmx_rmsnorm_bwd_blocks(int64_t T) { return std::min<int64_t>(std::max<int64_t>(1, hmx_ceil_div(T, 16)), 512); }

extern "C" int hmx_rmsnorm_bwd(const float* x, const float* scale, const float* rstd, const float* dy, int64_t T,
                               int64_t d, float* dx, float* dscale_part, void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x || !scale || !rstd || !dy || !dx || !dscale_part || d <= 0 || d > 256) return HMX_EINVAL;
    if (d % 4 == 0 && aligned16(x, scale, dy, dx))
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, dim3((int)hmx_rmsnorm_bwd_blocks(T)), dim3(256), 0, (hipStream_t)stream, x,
                           scale, rstd, dy, T, (int)d, dx, dscale_part);
    else
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<1>, dim3((int)hmx_rmsnorm_bwd_blocks(T)), dim3(256), 0, (hipStream_t)stream, x,
                           scale, rstd, dy, T, (int)d, dx, dscale_part);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" size_t hmx_colsum_workspace_bytes(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const int n = colsum_chunks(rows, cols);
    return n > 1 ? (size_t)n * (size_t)cols * sizeof(float) : 0;
}

extern "C" int hmx_colsum_f32(const float* P, int64_t rows, int64_t cols, int64_t ld, float* out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (cols == 0) return HMX_OK;
    if ((!P && rows > 0) || !out || rows < 0 || ld < cols) return HMX_EINVAL;   // no rows: the sums are zero
    const int n = colsum_chunks(rows, cols);
    if (n > 1 && workspace != nullptr) {
        if (workspace_bytes < hmx_colsum_workspace_bytes(rows, cols)) return HMX_EWORKSPACE;
        float* part = static_cast<float*>(workspace);
        const int64_t rpc = hmx_ceil_div(hmx_ceil_div(rows, n), 4) * 4;
        const int n_used = (int)hmx_ceil_div(rows, rpc);
        hipLaunchKernelGGL(colsum_chunks_kernel, dim3((unsigned)hmx_ceil_div(cols, 64), (unsigned)n_used), dim3(256), 0,
                           (hipStream_t)stream, P, rows, cols, ld, rpc, part);
        HMX_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)hmx_ceil_div(cols, 64)), dim3(1024), 0, (hipStream_t)stream,
                           part, n_used, cols, out);
        HMX_CHECK_LAUNCH();
        return HMX_OK;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)hmx_ceil_div(cols, 64)), dim3(1024), 0, (hipStream_t)stream, P, rows,
                       cols, ld, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

static int g_linear_split = 1;   // 1: split-bf16 (three MFMAs per product, fp32-class accuracy); 0: single bf16 MFMA

extern "C" int hmx_set_linear_precision(int split_bf16) {
    g_linear_split = split_bf16 != 0;
    return HMX_OK;
}

template <bool SPLIT>
static void launch_linear(int ks, dim3 grid, size_t lds, hipStream_t s, const LinearParams& p) {
    switch (ks) {
        case 1: hipLaunchKernelGGL((linear_kernel<1, SPLIT>), grid, dim3(256), lds, s, p); break;
        case 2: hipLaunchKernelGGL((linear_kernel<2, SPLIT>), grid, dim3(256), lds, s, p); break;
        case 4: hipLaunchKernelGGL((linear_kernel<4, SPLIT>), grid, dim3(256), lds, s, p); break;
        case 8: hipLaunchKernelGGL((linear_kernel<8, SPLIT>), grid, dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL((linear_kernel<16, SPLIT>), grid, dim3(256), lds, s, p); break;
    }
}

static int linear_launch(LinearParams& p, int n_z, hipStream_t stream) {
    const int64_t T = p.T, N = p.N, K = p.K;
    const int ks = ks_for((int)K);
    const int kp = ks * 16 + 8;
    const int copies = g_linear_split ? 2 : 1;
    int n_chunk = std::min<int>(256, (60 * 1024 / (kp * 2 * copies)) / 32 * 32);
    n_chunk = std::min<int>(n_chunk, (int)((N + 31) / 32 * 32));
    // A product over few tokens (the per-sequence projections of the fusion tail: B rows) is all weight staging: a
    // handful of workgroups would each convert the whole panel.  32 output columns per workgroup spread that work
    // (the tokens are re-read per panel, which costs nothing at this size).
    if (hmx_ceil_div(hmx_ceil_div(T, 32), 4) < 64) n_chunk = 32;
    p.n_chunk = n_chunk;
    const size_t lds = (size_t)n_chunk * kp * 2 * copies;
    // Every workgroup first stages its weight panel in LDS.  For K > 64 that panel is large enough that spreading the
    // token tiles evenly over one resident set of workgroups (~512 at this LDS footprint) beats one tile per wave
    // (measured: K = 128 projections 32.3 -> 27.6 us; for K <= 64 the extra waves hide more latency than the staging costs).
    const int64_t groups = hmx_ceil_div(hmx_ceil_div(T, 32), 4);
    const int64_t n_panels = hmx_ceil_div(N, n_chunk);
    int64_t gx = std::min<int64_t>(groups, 1024);
    if (ks >= 8) {
        const int64_t cap = std::max<int64_t>(64, 512 / n_panels);
        gx = hmx_ceil_div(groups, hmx_ceil_div(groups, cap));
    }
    const dim3 grid((unsigned)gx, (unsigned)n_panels, (unsigned)n_z);
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    if (g_linear_split) launch_linear<true>(ks, grid, lds, (hipStream_t)stream, p);
    else launch_linear<false>(ks, grid, lds, (hipStream_t)stream, p);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_linear_f32(const float* X, int64_t ldx, const float* W, int64_t w_rs, int64_t w_cs,
                              const float* bias, int64_t T, int64_t N, int64_t K, float* Y, int64_t ldy,
                              const float* R1, int64_t ldr1, float c1, const float* R2, int64_t ldr2, float c2,
                              const float* in_rowmask, const float* out_rowmask, int accumulate, void* stream) {
    HMX_ENTER();
    if (T == 0 || N == 0) return HMX_OK;
    if (!X || !W || !Y || K <= 0 || K > 256 || N < 0 || N > 65535) return HMX_EINVAL;
    LinearParams p{X, ldx, W, w_rs, w_cs, bias, T, (int)N, (int)K, 0, Y, ldy, R1, ldr1, c1, R2, ldr2, c2,
                   in_rowmask, out_rowmask, accumulate, 0, 0, 0, {}};
    return linear_launch(p, 1, (hipStream_t)stream);
}

// Y[:, h * y_head_stride + (0..N)] (+)= X[:, h * x_head_stride + (0..K)] Wv_h^T for h < n_heads in ONE launch; W: host
// array of n_heads device pointers, each viewed as Wv_h(n, k) = W[h][n * w_rs + k * w_cs].
extern "C" int hmx_heads_linear_f32(const float* X, int64_t ldx, int64_t x_head_stride, const float* const* W,
                                    int64_t w_rs, int64_t w_cs, int64_t T, int64_t N, int64_t K, float* Y, int64_t ldy,
                                    int64_t y_head_stride, int n_heads, int accumulate, void* stream) {
    HMX_ENTER();
    if (T == 0 || N == 0 || n_heads == 0) return HMX_OK;
    if (!X || !W || !Y || K <= 0 || K > 256 || N < 0 || N > 65535 || n_heads < 0 || n_heads > kLinearMaxHeads)
        return HMX_EINVAL;
    LinearParams p{X, ldx, nullptr, w_rs, w_cs, nullptr, T, (int)N, (int)K, 0, Y, ldy, nullptr, 0, 0.f, nullptr, 0, 0.f,
                   nullptr, nullptr, accumulate, n_heads, x_head_stride, y_head_stride, {}};
    for (int h = 0; h < n_heads; ++h) {
        if (!W[h]) return HMX_EINVAL;
        p.Wh[h] = W[h];
    }
    return linear_launch(p, n_heads, (hipStream_t)stream);
}

// Token splits of a weight-gradient launch: four per workgroup (one per wave).  Two workgroups fit a CU, so the chip
// runs 512 at a time, and the workgroups of a launch take about equally long: a grid of 528 would run as two rounds
// of which the second is nearly empty.  Hence as many groups as keep groups x tiles within ONE round (launches with
// more tiles than that run in several rounds of one group), at least 64 tokens per wave.
constexpr int64_t kWgradResident = 512, kWgradResidentSmall = 768;      // two / three workgroups per CU
static int64_t wgrad_groups(int64_t T_max, int64_t tiles, bool small = false) {
    int64_t g = std::max<int64_t>(1, (small ? kWgradResidentSmall : kWgradResident) / std::max<int64_t>(tiles, 1));
    g = std::min<int64_t>(g, hmx_ceil_div(T_max, 4 * 64));
    return std::max<int64_t>(1, std::min<int64_t>(g, 128));
}

static int g_wgrad_fp32 = 0;     // 0: split-bf16 operands on the bf16 MFMA (fp32-class accuracy); 1: fp32 MFMA

// Macro tile of a product: as many 32-wide tiles along N and K as vector loads allow (a lane reads tn consecutive
// columns of dY and tk of X, so base, row stride and extent must be multiples of the vector).
struct WgradShape {
    int tn, tk;
    int64_t tiles;
};
static WgradShape wgrad_shape(const hmx_wgrad_desc& d) {
    auto width = [](const float* base, int64_t ld, int64_t extent) {
        auto fits = [&](int v) { return reinterpret_cast<uintptr_t>(base) % (4 * v) == 0 && ld % v == 0 && extent % v == 0; };
        if (extent > 64 && fits(4)) return 4;
        if (extent > 32 && fits(2)) return 2;
        return 1;
    };
    WgradShape s{width(d.dY, d.ldy, d.N), width(d.X, d.ldx, d.K), 0};
    // accumulator tiles per wave: eight on the fp32 MFMA (its multiply rate is the bound: fewest operand re-reads),
    // four on the bf16 MFMA (bound by the reads in flight: smaller tiles leave registers for them; measured
    // 1.34 -> 1.11 ms per configs[1] step)
    const int cap = g_wgrad_fp32 ? 8 : 4;
    while (s.tn * s.tk > cap) {
        if (s.tk >= s.tn) s.tk /= 2;
        else s.tn /= 2;
    }
    s.tiles = hmx_ceil_div(d.N, 32 * s.tn) * hmx_ceil_div(d.K, 32 * s.tk);
    return s;
}
static bool wgrad_groups_ok(const hmx_wgrad_desc& d) {
    auto ok = [](int64_t extent, int32_t g, int32_t gp) {
        if (g == 0 && gp == 0) return true;
        return g > 0 && gp >= g && extent % gp == 0;
    };
    return ok(d.N, d.n_group, d.n_group_padded) && ok(d.K, d.k_group, d.k_group_padded);
}

// The counters of a launch: the slot that belongs to the launch's (device, stream).  Launches on one stream run one
// after the other, so they can share a slot (each leaves its counters at zero); launches on different streams of a
// GPU may overlap and must not.  Sixteen streams per GPU hold a slot at a time; a seventeenth takes over the slot
// whose stream launched longest ago.
static int wgrad_tickets_for_call(hipStream_t stream, unsigned** out) {
    struct Slots {
        unsigned* base = nullptr;
        hipStream_t owner[kWgradSlots] = {};
        uint64_t last_use[kWgradSlots] = {};
        int used = 0;
    };
    static Slots per_device[64];          // each GPU has its own copy of the array
    static std::mutex mu;
    static uint64_t clock = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return HMX_ELAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    Slots& sl = per_device[dev];
    if (sl.base == nullptr) {
        unsigned* tickets = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&tickets), HIP_SYMBOL(g_wgrad_tickets)) != hipSuccess || !tickets)
            return HMX_ELAUNCH;
        sl.base = tickets;
    }
    int slot = -1;
    for (int i = 0; i < sl.used; ++i)
        if (sl.owner[i] == stream) slot = i;
    if (slot < 0) {
        if (sl.used < kWgradSlots) {
            slot = sl.used++;
        } else {
            slot = 0;
            for (int i = 1; i < kWgradSlots; ++i)
                if (sl.last_use[i] < sl.last_use[slot]) slot = i;
        }
        sl.owner[slot] = stream;
    }
    sl.last_use[slot] = ++clock;
    *out = sl.base + (size_t)slot * kWgradMaxTiles;
    return HMX_OK;
}

// Walks the descriptors of a batch: macro tiles and slab floats (per token-split group) of the whole launch.
struct WgradPlan {
    int64_t tiles = 0, slab_floats = 0, t_max = 1;
    bool small = true;       // no job with more than four accumulator tiles per wave
};
static WgradPlan wgrad_plan(const hmx_wgrad_desc* descs, int64_t n) {
    WgradPlan pl;
    for (int64_t j = 0; j < n; ++j) {
        if (descs[j].N <= 0 || descs[j].K <= 0) continue;
        const WgradShape sh = wgrad_shape(descs[j]);
        pl.tiles += sh.tiles;
        pl.slab_floats += sh.tiles * wgrad_slab_floats(sh.tn, sh.tk);
        pl.t_max = std::max<int64_t>(pl.t_max, descs[j].T);
        pl.small = pl.small && sh.tn * sh.tk <= 4;
    }
    // three workgroups per CU pay when a wave has many trips to run (configs[4]: 5.9 -> 4.85 ms); with few, the longer
    // chain of slabs the last workgroup of a tile adds costs more than the extra waves bring (configs[1]: 1.11 -> 1.42 ms)
    pl.small = pl.small && !g_wgrad_fp32 && pl.tiles > 0 &&
               pl.t_max / (4 * wgrad_groups(pl.t_max, pl.tiles, true)) >= 2048;
    return pl;
}

extern "C" int hmx_set_wgrad_precision(int fp32_mfma) {
    g_wgrad_fp32 = fp32_mfma != 0;
    return HMX_OK;
}

extern "C" size_t hmx_linear_wgrad_batch_workspace_bytes(const hmx_wgrad_desc* descs, int64_t n) {
    if (!descs || n <= 0) return 256;
    const WgradPlan pl = wgrad_plan(descs, n);
    if (pl.tiles == 0) return 256;
    return (size_t)(wgrad_groups(pl.t_max, pl.tiles, pl.small) * pl.slab_floats) * sizeof(float) + 256;
}

extern "C" int hmx_linear_wgrad_batch_f32(const hmx_wgrad_desc* descs, int64_t n, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    HMX_ENTER();
    if (n == 0) return HMX_OK;
    if (!descs || n < 0 || n > kWgradMaxJobs || !workspace) return HMX_EINVAL;
    if (workspace_bytes < hmx_linear_wgrad_batch_workspace_bytes(descs, n)) return HMX_EWORKSPACE;
    const WgradPlan pl = wgrad_plan(descs, n);
    if (pl.tiles > kWgradMaxTiles) return HMX_EINVAL;
    const int64_t G = wgrad_groups(pl.t_max, std::max<int64_t>(pl.tiles, 1), pl.small);
    WgradParams p{};
    int64_t tiles = 0, slab0 = 0;
    int nj = 0;
    for (int64_t j = 0; j < n; ++j) {
        const hmx_wgrad_desc& d = descs[j];
        if (d.N == 0 || d.K == 0) continue;
        if (!d.dY || !d.X || !d.dW || d.T <= 0 || d.N < 0 || d.K < 0 || !wgrad_groups_ok(d)) return HMX_EINVAL;
        const WgradShape sh = wgrad_shape(d);
        WgradJob& jb = p.job[nj++];
        jb.dY = d.dY; jb.X = d.X; jb.in_rowmask = d.in_rowmask; jb.dW = d.dW; jb.dbias = d.dbias;
        jb.ldy = d.ldy; jb.ldx = d.ldx; jb.ldw = d.ldw; jb.T = d.T;
        jb.N = (int)d.N; jb.K = (int)d.K; jb.with_bias = d.dbias != nullptr; jb.accumulate = d.accumulate != 0;
        jb.tn = sh.tn; jb.tk = sh.tk;
        jb.n_grp = d.n_group; jb.n_grpp = d.n_group_padded; jb.k_grp = d.k_group; jb.k_grpp = d.k_group_padded;
        jb.tile0 = (int)tiles;
        jb.slab0 = slab0;
        tiles += sh.tiles;
        slab0 += G * sh.tiles * wgrad_slab_floats(sh.tn, sh.tk);
    }
    if (nj == 0) return HMX_OK;
    p.n_jobs = nj;
    p.fp32_mfma = g_wgrad_fp32;
    p.part = static_cast<float*>(workspace);
    const int rc = wgrad_tickets_for_call((hipStream_t)stream, &p.tickets);
    if (rc != HMX_OK) return rc;
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    if (pl.small) hipLaunchKernelGGL(wgrad_kernel<true>, dim3((unsigned)G, (unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_kernel<false>, dim3((unsigned)G, (unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" size_t hmx_linear_wgrad_workspace_bytes(int64_t T, int64_t N, int64_t K) {
    if (T <= 0 || N <= 0 || K <= 0) return 256;
    int64_t floats = 0;                         // the operands' alignment picks the macro tile: enough for any of them
    for (int tn = 1; tn <= 4; tn *= 2)
        for (int tk = 1; tk * tn <= 8 && tk <= 4; tk *= 2) {
            const int64_t tiles = hmx_ceil_div(N, 32 * tn) * hmx_ceil_div(K, 32 * tk);
            floats = std::max<int64_t>(floats, wgrad_groups(T, tiles, tn * tk <= 4 && !g_wgrad_fp32) * tiles * wgrad_slab_floats(tn, tk));
        }
    return (size_t)floats * sizeof(float) + 256;
}

extern "C" int hmx_linear_wgrad_f32(const float* dY, int64_t ldy, const float* X, int64_t ldx, const float* in_rowmask,
                                    int64_t T, int64_t N, int64_t K, float* dW, int64_t ldw, float* dbias,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0 || K == 0) return HMX_OK;
    if (!dY || !X || !dW || !workspace || T <= 0) return HMX_EINVAL;
    if (workspace_bytes < hmx_linear_wgrad_workspace_bytes(T, N, K)) return HMX_EWORKSPACE;
    hmx_wgrad_desc d{dY, ldy, X, ldx, in_rowmask, T, N, K, dW, ldw, dbias, 0, 0, 0, 0, 0};
    return hmx_linear_wgrad_batch_f32(&d, 1, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------- plain linear recurrence
// h[t] = a[t] * h[t-1] + x[t] over (B, L, C): one thread per (b, c) chain, consecutive c in consecutive lanes so
// every step is one coalesced row access; four steps of loads are issued ahead of the dependent multiply-adds.
__global__ __launch_bounds__(256) void pscan_fwd_kernel(const float* __restrict__ A, const float* __restrict__ X,
                                                        int64_t B, int L, int64_t C, float* __restrict__ H) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int64_t b = idx / C, c = idx % C;
    const int64_t base = b * L * C + c;
    float h = 0.f;
    int t = 0;
    for (; t + 4 <= L; t += 4) {
        float a[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = A[base + (int64_t)(t + u) * C];
            x[u] = X[base + (int64_t)(t + u) * C];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = a[u] * h + x[u];
            H[base + (int64_t)(t + u) * C] = h;
        }
    }
    for (; t < L; ++t) {
        h = A[base + (int64_t)t * C] * h + X[base + (int64_t)t * C];
        H[base + (int64_t)t * C] = h;
    }
}

// g[t] = dH[t] + a[t+1] * g[t+1];  dX[t] = g[t];  dA[t] = g[t] * h[t-1]  (h[-1] = 0)
__global__ __launch_bounds__(256) void pscan_bwd_kernel(const float* __restrict__ A, const float* __restrict__ H,
                                                        const float* __restrict__ dH, int64_t B, int L, int64_t C,
                                                        float* __restrict__ dA, float* __restrict__ dX) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int64_t b = idx / C, c = idx % C;
    const int64_t base = b * L * C + c;
    float g = 0.f, a_next = 0.f;
    for (int t = L - 1; t >= 0; --t) {
        const int64_t o = base + (int64_t)t * C;
        g = dH[o] + a_next * g;
        const float h_prev = t > 0 ? H[o - C] : 0.f;
        dX[o] = g;
        dA[o] = g * h_prev;
        a_next = A[o];
    }
}

// ---------------------------------------------------------------------------- one-query attention
// The fusion tail of the multi-scale encoder (hydra3.py:528-533): per sequence ONE query attends over the L
// positions of the long-term representation.  kv rows are [k (d) | v (d)] per token, heads are column blocks of
// hd = d / H.  One wave per (sequence, head): lanes own positions for the scores and the softmax, then own
// feature columns for the weighted sums (weights parked in LDS).  No key-padding mask (the reference passes none).
// drop (B, H, L), optional: multiplier of the attention weights (0 or 1 / (1 - p)), i.e. the module's dropout.
constexpr int kAttnMaxL = 4096;

__global__ __launch_bounds__(64) void attn1q_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                        const float* __restrict__ drop, int L, int d, int H,
                                                        float* __restrict__ p_out, float* __restrict__ out) {
    __shared__ float w_sh[kAttnMaxL];
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* qh = q + (int64_t)b * d + h * hd;
    const float* kvb = kv + (int64_t)b * L * 2 * d;
    float mx = -INFINITY;
    for (int l = lane; l < L; l += 64) {
        const float* kr = kvb + (int64_t)l * 2 * d + h * hd;
        float s = 0.f;
        for (int j = 0; j < hd; ++j) s = __builtin_fmaf(qh[j], kr[j], s);
        s *= scale;
        w_sh[l] = s;
        mx = fmaxf(mx, s);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int l = lane; l < L; l += 64) {
        const float e = expf(w_sh[l] - mx);
        w_sh[l] = e;
        sum += e;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
    for (int l = lane; l < L; l += 64) {
        const float pr = w_sh[l] * inv;
        p_out[((int64_t)b * H + h) * L + l] = pr;
        w_sh[l] = drop ? pr * drop[((int64_t)b * H + h) * L + l] : pr;
    }
    __syncthreads();
    for (int j = lane; j < hd; j += 64) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = __builtin_fmaf(w_sh[l], kvb[(int64_t)l * 2 * d + d + h * hd + j], acc);
        out[(int64_t)b * d + h * hd + j] = acc;
    }
}

__global__ __launch_bounds__(64) void attn1q_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                        const float* __restrict__ drop, const float* __restrict__ p_in,
                                                        const float* __restrict__ dout, int L, int d, int H,
                                                        float* __restrict__ dq, float* __restrict__ dkv) {
    __shared__ float ds_sh[kAttnMaxL];   // d score
    __shared__ float wv_sh[kAttnMaxL];   // weight that multiplied v (p * drop)
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* qh = q + (int64_t)b * d + h * hd;
    const float* doh = dout + (int64_t)b * d + h * hd;
    const float* kvb = kv + (int64_t)b * L * 2 * d;
    float* dkvb = dkv + (int64_t)b * L * 2 * d;
    const float* pb = p_in + ((int64_t)b * H + h) * L;
    const float* db = drop ? drop + ((int64_t)b * H + h) * L : nullptr;
    // dw[l] = drop[l] * <dout, v_l>;  dot = sum_l p[l] dw[l];  ds[l] = p[l] (dw[l] - dot)
    float dot = 0.f;
    for (int l = lane; l < L; l += 64) {
        const float* vr = kvb + (int64_t)l * 2 * d + d + h * hd;
        float dw = 0.f;
        for (int j = 0; j < hd; ++j) dw = __builtin_fmaf(doh[j], vr[j], dw);
        const float m = db ? db[l] : 1.0f;
        dw *= m;
        ds_sh[l] = dw;
        wv_sh[l] = pb[l] * m;
        dot += pb[l] * dw;
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    for (int l = lane; l < L; l += 64) ds_sh[l] = pb[l] * (ds_sh[l] - dot) * scale;   // includes the 1/sqrt(hd)
    __syncthreads();
    for (int j = lane; j < hd; j += 64) {
        float acc = 0.f;
        const float qj = qh[j], dj = doh[j];
        for (int l = 0; l < L; ++l) {
            const int64_t o = (int64_t)l * 2 * d + h * hd + j;
            acc = __builtin_fmaf(ds_sh[l], kvb[o], acc);
            dkvb[o] = ds_sh[l] * qj;          // d k
            dkvb[o + d] = wv_sh[l] * dj;      // d v
        }
        dq[(int64_t)b * d + h * hd + j] = acc;
    }
}

// Row-parallel form of the two kernels above: one workgroup per sequence, every k / v row read (and every dk / dv row
// written) 16 bytes per lane by next_pow2(d / 4) neighbouring lanes, four rows in flight per lane; the per-head dot
// products are finished with shuffles inside the head's lanes, the weighted sums over positions by row groups whose
// partial sums are added in group order.  Needs head widths whose quarter is a power of two (hd = 32, 64, ...) and
// H * L <= kAttnRowsMaxHL; other shapes use the kernels above.  (One wave per (sequence, head) with a lane per
// position read its rows four bytes at a time: 39 + 61 us per step at B = 1024, L = 100, d = 64.)
constexpr int kAttnRowsMaxHL = 8192;

__device__ __forceinline__ float head_lanes_sum(float v, int lanes) {      // lanes: power of two <= 64, aligned groups
    for (int off = lanes >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// scores s_sh[h * L + l] = scale_or_1 * <vec_h, row_l[col0 + h * hd ...]> for all l, h (vec: q or dout; rows: k or v)
__device__ __forceinline__ void attn_rows_dots(const float4* __restrict__ rows4, int row_f4, int col0_f4, const float4 vq,
                                               int L, int cpr, int lpr, int hl, float mul, float* __restrict__ s_sh) {
    const int c = threadIdx.x % lpr, g = threadIdx.x / lpr, rp = 256 / lpr;
    const bool c_ok = c < cpr;
    const int h = c / hl;
    for (int l0 = 0; l0 < L; l0 += 4 * rp) {
        float4 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * rp + g;
            kr[u] = (c_ok && l < L) ? rows4[(int64_t)l * row_f4 + col0_f4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * rp + g;
            float part = kr[u].x * vq.x + kr[u].y * vq.y + kr[u].z * vq.z + kr[u].w * vq.w;
            part = head_lanes_sum(part, hl);
            if (c_ok && l < L && (c % hl) == 0) s_sh[h * L + l] = part * mul;
        }
    }
}

// acc4 = sum_l w_sh[h * L + l] * rows_l[col0 + 4 c ...] for this lane's column chunk, rows dealt to the row groups and the
// groups' partial sums added in group order; valid in the threads of group 0
__device__ __forceinline__ float4 attn_rows_weighted_sum(const float4* __restrict__ rows4, int row_f4, int col0_f4, int L,
                                                         int cpr, int lpr, int hl, const float* __restrict__ w_sh,
                                                         float4* __restrict__ red /* [256] */) {
    const int c = threadIdx.x % lpr, g = threadIdx.x / lpr, rp = 256 / lpr;
    const bool c_ok = c < cpr;
    const int h = c_ok ? c / hl : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l0 = g; l0 < L; l0 += 4 * rp) {
        float4 vr[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * rp;
            const bool ok = c_ok && l < L;
            vr[u] = ok ? rows4[(int64_t)l * row_f4 + col0_f4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            w[u] = ok ? w_sh[h * L + l] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc.x = __builtin_fmaf(w[u], vr[u].x, acc.x);
            acc.y = __builtin_fmaf(w[u], vr[u].y, acc.y);
            acc.z = __builtin_fmaf(w[u], vr[u].z, acc.z);
            acc.w = __builtin_fmaf(w[u], vr[u].w, acc.w);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g == 0) {
        for (int gg = 0; gg < rp; ++gg) {
            const float4 o = red[gg * lpr + c];
            tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w;
        }
    }
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(256) void attn1q_rows_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                              const float* __restrict__ drop, int L, int d, int H,
                                                              int lpr, float* __restrict__ p_out, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char attn_smem[];
    float* s_sh = reinterpret_cast<float*>(attn_smem);                       // [H][L]
    float4* red = reinterpret_cast<float4*>(attn_smem + (size_t)((H * L + 3) / 4 * 4) * sizeof(float));   // [256]
    const int b = blockIdx.x, cpr = d / 4, hd = d / H, hl = hd / 4;
    const int c = threadIdx.x % lpr, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = 1.0f / sqrtf((float)hd);
    const float4* kv4 = reinterpret_cast<const float4*>(kv) + (int64_t)b * L * 2 * cpr;
    const float4 q4 = c < cpr ? reinterpret_cast<const float4*>(q)[(int64_t)b * cpr + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    attn_rows_dots(kv4, 2 * cpr, 0, q4, L, cpr, lpr, hl, scale, s_sh);
    __syncthreads();
    for (int h = wave; h < H; h += 4) {                                      // softmax of a head: one wave
        float* sh = s_sh + h * L;
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, sh[l]);
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float e = expf(sh[l] - mx);
            sh[l] = e;
            sum += e;
        }
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float inv = 1.0f / sum;
        for (int l = lane; l < L; l += 64) {
            const float pr = sh[l] * inv;
            p_out[((int64_t)b * H + h) * L + l] = pr;
            sh[l] = drop ? pr * drop[((int64_t)b * H + h) * L + l] : pr;
        }
    }
    __syncthreads();
    const float4 o4 = attn_rows_weighted_sum(kv4, 2 * cpr, cpr, L, cpr, lpr, hl, s_sh, red);
    if (threadIdx.x < lpr && c < cpr) reinterpret_cast<float4*>(out)[(int64_t)b * cpr + c] = o4;
}

__global__ __launch_bounds__(256) void attn1q_rows_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                              const float* __restrict__ drop, const float* __restrict__ p_in,
                                                              const float* __restrict__ dout, int L, int d, int H, int lpr,
                                                              float* __restrict__ dq, float* __restrict__ dkv) {
    extern __shared__ __attribute__((aligned(16))) char attn_smem[];
    const int hl_pad = (H * L + 3) / 4 * 4;
    float* ds_sh = reinterpret_cast<float*>(attn_smem);                      // [H][L] d score
    float* wv_sh = ds_sh + hl_pad;                                           // [H][L] weight that multiplied v (p * drop)
    float4* red = reinterpret_cast<float4*>(wv_sh + hl_pad);                 // [256]
    const int b = blockIdx.x, cpr = d / 4, hd = d / H, hl = hd / 4;
    const int c = threadIdx.x % lpr, g = threadIdx.x / lpr, rp = 256 / lpr, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = 1.0f / sqrtf((float)hd);
    const float4* kv4 = reinterpret_cast<const float4*>(kv) + (int64_t)b * L * 2 * cpr;
    float4* dkv4 = reinterpret_cast<float4*>(dkv) + (int64_t)b * L * 2 * cpr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 q4 = c < cpr ? reinterpret_cast<const float4*>(q)[(int64_t)b * cpr + c] : zero4;
    const float4 do4 = c < cpr ? reinterpret_cast<const float4*>(dout)[(int64_t)b * cpr + c] : zero4;
    // dw[l] = drop[l] * <dout, v_l>;  dot = sum_l p[l] dw[l];  ds[l] = p[l] (dw[l] - dot) / sqrt(hd)
    attn_rows_dots(kv4, 2 * cpr, cpr, do4, L, cpr, lpr, hl, 1.0f, ds_sh);
    __syncthreads();
    for (int h = wave; h < H; h += 4) {
        const float* pb = p_in + ((int64_t)b * H + h) * L;
        const float* db = drop ? drop + ((int64_t)b * H + h) * L : nullptr;
        float dot = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float m = db ? db[l] : 1.0f;
            const float dw = ds_sh[h * L + l] * m;
            ds_sh[h * L + l] = dw;
            wv_sh[h * L + l] = pb[l] * m;
            dot += pb[l] * dw;
        }
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        for (int l = lane; l < L; l += 64) ds_sh[h * L + l] = pb[l] * (ds_sh[h * L + l] - dot) * scale;
    }
    __syncthreads();
    const float4 dq4 = attn_rows_weighted_sum(kv4, 2 * cpr, 0, L, cpr, lpr, hl, ds_sh, red);
    if (threadIdx.x < lpr && c < cpr) reinterpret_cast<float4*>(dq)[(int64_t)b * cpr + c] = dq4;
    if (c < cpr) {
        const int h = c / hl;
        for (int l = g; l < L; l += rp) {
            const float a = ds_sh[h * L + l], w = wv_sh[h * L + l];
            dkv4[(int64_t)l * 2 * cpr + c] = make_float4(a * q4.x, a * q4.y, a * q4.z, a * q4.w);              // d k
            dkv4[(int64_t)l * 2 * cpr + cpr + c] = make_float4(w * do4.x, w * do4.y, w * do4.z, w * do4.w);    // d v
        }
    }
}

// lanes per row of the row-parallel form, or 0 when the shape takes the wave-per-head kernels
static int attn_rows_lpr(int64_t L, int64_t d, int64_t H, const void* a, const void* b, const void* c2, const void* e) {
    if (d % 4 || d % H || d > 256 || H * L > kAttnRowsMaxHL) return 0;
    const int64_t hd = d / H;
    if (hd % 4) return 0;
    const int64_t hl = hd / 4;
    if (hl & (hl - 1)) return 0;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c2) |
         reinterpret_cast<uintptr_t>(e)) & 15) return 0;
    int lpr = 1;
    while (lpr < d / 4) lpr *= 2;
    return lpr;
}

extern "C" int hmx_attn1q_fwd(const float* q, const float* kv, const float* drop, int64_t B, int64_t L, int64_t d,
                              int64_t H, float* p_out, float* out, void* stream) {
    HMX_ENTER();
    if (B == 0) return HMX_OK;
    if (!q || !kv || !p_out || !out || L <= 0 || L > kAttnMaxL || d <= 0 || H <= 0 || d % H) return HMX_EINVAL;
    if (const int lpr = attn_rows_lpr(L, d, H, q, kv, out, nullptr)) {
        const size_t lds = (size_t)((H * L + 3) / 4 * 4) * sizeof(float) + 256 * sizeof(float4);
        hipLaunchKernelGGL(attn1q_rows_fwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, q, kv, drop, (int)L,
                           (int)d, (int)H, lpr, p_out, out);
    } else {
        hipLaunchKernelGGL(attn1q_fwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, (hipStream_t)stream, q, kv, drop, (int)L,
                           (int)d, (int)H, p_out, out);
    }
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_attn1q_bwd(const float* q, const float* kv, const float* drop, const float* p, const float* dout,
                              int64_t B, int64_t L, int64_t d, int64_t H, float* dq, float* dkv, void* stream) {
    HMX_ENTER();
    if (B == 0) return HMX_OK;
    if (!q || !kv || !p || !dout || !dq || !dkv || L <= 0 || L > kAttnMaxL || d <= 0 || H <= 0 || d % H)
        return HMX_EINVAL;
    if (const int lpr = attn_rows_lpr(L, d, H, q, kv, dout, dkv); lpr != 0 && (reinterpret_cast<uintptr_t>(dq) & 15) == 0) {
        const size_t lds = 2 * (size_t)((H * L + 3) / 4 * 4) * sizeof(float) + 256 * sizeof(float4);
        hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&attn1q_rows_bwd_kernel), 2 * kAttnRowsMaxHL * 4 + 4096 + 64);
        hipLaunchKernelGGL(attn1q_rows_bwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, q, kv, drop, p, dout,
                           (int)L, (int)d, (int)H, lpr, dq, dkv);
    } else {
        hipLaunchKernelGGL(attn1q_bwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, (hipStream_t)stream, q, kv, drop, p, dout,
                           (int)L, (int)d, (int)H, dq, dkv);
    }
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_pscan_fwd(const float* A, const float* X, int64_t B, int64_t L, int64_t C, float* H, void* stream) {
    HMX_ENTER();
    if (B == 0 || L == 0 || C == 0) return HMX_OK;
    if (!A || !X || !H || B < 0 || L < 0 || C < 0 || L >= ((int64_t)1 << 31)) return HMX_EINVAL;
    hipLaunchKernelGGL(pscan_fwd_kernel, dim3((unsigned)hmx_ceil_div(B * C, 256)), dim3(256), 0, (hipStream_t)stream, A, X,
                       B, (int)L, C, H);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_pscan_bwd(const float* A, const float* H, const float* dH, int64_t B, int64_t L, int64_t C,
                             float* dA, float* dX, void* stream) {
    HMX_ENTER();
    if (B == 0 || L == 0 || C == 0) return HMX_OK;
    if (!A || !H || !dH || !dA || !dX || B < 0 || L < 0 || C < 0 || L >= ((int64_t)1 << 31)) return HMX_EINVAL;
    hipLaunchKernelGGL(pscan_bwd_kernel, dim3((unsigned)hmx_ceil_div(B * C, 256)), dim3(256), 0, (hipStream_t)stream, A, H,
                       dH, B, (int)L, C, dA, dX);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_silu_rope_fwd(const float* x1, const float* cos_t, const float* sin_t, int64_t T, int64_t L,
                                 int64_t d, int64_t hd, float* s, float* y2, void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x1 || !cos_t || !sin_t || !s || !y2 || d % 2 || hd % 2 || d % hd || L <= 0) return HMX_EINVAL;
    hipLaunchKernelGGL(silu_rope_fwd_kernel, dim3(grid_for(T * d / 2, 256)), dim3(256), 0, (hipStream_t)stream, x1, cos_t,
                       sin_t, T, (int)L, (int)d, (int)hd, s, y2);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_silu_rope_bwd(const float* x1, const float* cos_t, const float* sin_t, const float* ds,
                                 const float* dy2, int64_t T, int64_t L, int64_t d, int64_t hd, float* dx1,
                                 void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x1 || !cos_t || !sin_t || !ds || !dy2 || !dx1 || d % 2 || hd % 2 || d % hd || L <= 0) return HMX_EINVAL;
    hipLaunchKernelGGL(silu_rope_bwd_kernel, dim3(grid_for(T * d / 2, 256)), dim3(256), 0, (hipStream_t)stream, x1, cos_t,
                       sin_t, ds, dy2, T, (int)L, (int)d, (int)hd, dx1);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_binary_fwd(const float* a, const float* b, int64_t n, float c, int mode, float* out, void* stream) {
    HMX_ENTER();
    if (n == 0) return HMX_OK;
    if (!a || !b || !out || mode < 0 || mode > 1) return HMX_EINVAL;
    if (n % 4 == 0 && aligned16(a, b, out))
        hipLaunchKernelGGL(binary_fwd_vec4_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), n / 4, c, mode,
                           reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(binary_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, c, mode, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_binary_bwd(const float* a, const float* b, const float* g, int64_t n, float c, int mode, float* da,
                              float* db, void* stream) {
    HMX_ENTER();
    if (n == 0) return HMX_OK;
    if (!a || !b || !g || !da || !db || mode < 0 || mode > 1) return HMX_EINVAL;
    if (n % 4 == 0 && aligned16(a, b, g, da, db))
        hipLaunchKernelGGL(binary_bwd_vec4_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                           reinterpret_cast<const float4*>(g), n / 4, c, mode, reinterpret_cast<float4*>(da),
                           reinterpret_cast<float4*>(db));
    else
        hipLaunchKernelGGL(binary_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, g, n, c, mode,
                           da, db);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_combine_f32(const float* x0, const float* x1, const float* x2, const float* x3, float c0, float c1,
                               float c2, float c3, const float* rowmask, int64_t T, int64_t d, float* out,
                               void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x0 || !out || d <= 0) return HMX_EINVAL;
    if (d % 4 == 0 && aligned16(x0, x1, x2, x3, out))
        hipLaunchKernelGGL(combine_vec4_kernel, dim3(grid_for(T * d / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(x0), reinterpret_cast<const float4*>(x1),
                           reinterpret_cast<const float4*>(x2), reinterpret_cast<const float4*>(x3), c0, c1, c2, c3, rowmask, T,
                           (int)(d / 4), reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(combine_kernel, dim3(grid_for(T * d, 256)), dim3(256), 0, (hipStream_t)stream, x0, x1, x2, x3, c0,
                           c1, c2, c3, rowmask, T, (int)d, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_seq_index(const int64_t* ids, const int64_t* ts, int64_t T, int64_t L, int64_t time_span, float* mask,
                             int64_t* pos_idx, int64_t* time_bin, void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!ids || !ts || !mask || !pos_idx || !time_bin || L <= 0 || T % L) return HMX_EINVAL;
    hipLaunchKernelGGL(seq_index_kernel, dim3(grid_for(T, 256)), dim3(256), 0, (hipStream_t)stream, ids, ts, T, (int)L,
                       (int)time_span, mask, pos_idx, time_bin);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_conv_silu_fwd(const float* x, const float* w, const float* b, int64_t T, int64_t L, int64_t C,
                                 float* xc, void* stream) {
    HMX_ENTER();
    if (T == 0) return HMX_OK;
    if (!x || !w || !b || !xc || L <= 0 || T % L || C <= 0) return HMX_EINVAL;
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                                     reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(xc)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(conv_silu_fwd_vec4_kernel,
                           dim3(grid_for(T / L * hmx_ceil_div(L, kConvRun) * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                           x, w, b, T, (int)L, (int)C, xc);
    else
        hipLaunchKernelGGL(conv_silu_fwd_kernel, dim3(grid_for(T * C, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, T,
                           (int)L, (int)C, xc);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_conv_silu_bwd(const float* x, const float* w, const float* b, const float* dxc, int64_t B, int64_t L,
                                 int64_t C, float* dx, float* part, void* stream) {
    HMX_ENTER();
    if (B == 0) return HMX_OK;
    if (!x || !w || !b || !dxc || !dx || !part || L <= 0 || C <= 0) return HMX_EINVAL;
    hipLaunchKernelGGL(conv_silu_bwd_kernel, dim3((unsigned)B, (unsigned)hmx_ceil_div(C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, w, b, dxc, (int)L, (int)C, dx, part);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

static bool scan_shape_ok(int64_t C, int64_t ED, int64_t H) {
    return ED >= 32 && (ED & (ED - 1)) == 0 && C == ED * H && H * kP * kScanChunk * 4 <= 64 * 1024;
}

extern "C" int64_t hmx_scan_state_floats(int64_t B, int64_t L, int64_t C) {
    if (B <= 0 || L <= 0 || C <= 0) return 0;
    return B * (int64_t)scan_n_ck((int)L) * C * kN;
}

extern "C" int hmx_scan_fwd(const float* xc, const float* xz, const float* dbc, const float* Wdt, const float* bdt,
                            const float* A_log, const float* D, int64_t B, int64_t L, int64_t C, int64_t ED, int64_t H,
                            float* yg, float* hs, void* stream) {
    HMX_ENTER();
    if (B == 0) return HMX_OK;
    if (!xc || !dbc || !Wdt || !bdt || !A_log || !D || !yg || L <= 0 || !scan_shape_ok(C, ED, H)) return HMX_EINVAL;
    ScanParams p{xc, xz, dbc, Wdt, bdt, A_log, D, (int)B, (int)L, (int)C, (int)ED, (int)H, yg, hs};
    void* span = hmx_profile_begin("scan_fwd", (hipStream_t)stream);
    hipLaunchKernelGGL(scan_fwd_kernel, dim3((unsigned)B, (unsigned)hmx_ceil_div(C, 256)), dim3(256),
                       (size_t)kScanChunk * H * kP * 4, (hipStream_t)stream, p);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int64_t hmx_scan_groups_per_head(int64_t ED) { return ED >= 64 ? ED / 64 : 1; }

extern "C" int hmx_scan_bwd(const float* xc, const float* xz, const float* dbc, const float* Wdt, const float* bdt,
                            const float* A_log, const float* D, const float* hs, const float* dyg, int64_t B, int64_t L,
                            int64_t C, int64_t ED, int64_t H, float* dxc, float* dxz, float* ddbc_part,
                            float* par_part, void* stream) {
    HMX_ENTER();
    if (B == 0) return HMX_OK;
    if (!xc || !dbc || !Wdt || !bdt || !A_log || !D || (!hs && scan_n_ck((int)L) > 0) || !dyg || !dxc || (xz && !dxz) ||
        !ddbc_part || !par_part || L <= 0 || !scan_shape_ok(C, ED, H))
        return HMX_EINVAL;
    ScanBwdParams q{};
    q.f = ScanParams{xc, xz, dbc, Wdt, bdt, A_log, D, (int)B, (int)L, (int)C, (int)ED, (int)H, nullptr,
                     const_cast<float*>(hs)};
    q.dyg = dyg;
    q.dxc = dxc;
    q.dxz = dxz;
    q.ddbc_part = ddbc_part;
    q.par_part = par_part;
    q.GPH = (int)hmx_scan_groups_per_head(ED);
    const dim3 grid((unsigned)B, (unsigned)hmx_ceil_div(C, 256));
    const size_t lds = (size_t)kCk * 4 * 256 * sizeof(float4) + (size_t)kScanChunkBwd * H * kP * 4;
    hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&scan_bwd_kernel<64>), 160 * 1024);
    hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&scan_bwd_kernel<32>), 160 * 1024);
    if (lds > 160 * 1024) return HMX_EINVAL;
    void* span = hmx_profile_begin("scan_bwd", (hipStream_t)stream);
    if (ED >= 64) hipLaunchKernelGGL(scan_bwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(scan_bwd_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, q);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_sum_groups_f32(const float* part, int64_t rows, int64_t groups, int64_t width, float* out,
                                  void* stream) {
    HMX_ENTER();
    if (rows == 0) return HMX_OK;
    if (!part || !out || groups <= 0 || width <= 0) return HMX_EINVAL;
    hipLaunchKernelGGL(sum_groups_kernel, dim3(grid_for(rows * width, 256)), dim3(256), 0, (hipStream_t)stream, part, rows,
                       (int)groups, (int)width, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
