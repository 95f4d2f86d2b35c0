// Fused token-parallel halves of a Hydra block (hydra3.py:359-389, 410-457), forward and backward.
//
// A block is   x -> [dropout, RMSNorm1, linear1, SiLU, RoPE, in_proj]  ->  conv / deltaBC / scan (sequence kernels)
//                -> [out_proj, y1*y2, linear2 + residuals, RMSNorm2, SwiGLU FFN + residual, mask]  -> out.
// The two bracketed halves have no dependence between tokens, so each is ONE kernel here: a wave owns 32 tokens and
// carries every intermediate in the accumulator layout of the 32x32 MFMA with FEATURES on the rows and TOKENS on the
// columns (lane & 31 = token, the 16 registers x 2 lane halves = 32 features).  In that layout
//   * a pointwise op (SiLU, RoPE pairs, gates, residuals) is lane-local,
//   * a row statistic (RMSNorm) is a sum over a lane's registers plus ONE exchange with lane ^ 32,
//   * a tile is already the B operand of the next product (k order permuted; the weights are packed to match),
// so no intermediate goes through LDS or HBM.  Weights arrive pre-packed (hmx_block_pack_f32) as bf16 hi / lo
// fragment pairs in exactly the order a lane consumes them: the A operand of an MFMA is one 16-byte load per lane from
// L2.  Products are split-bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate): fp32-class accuracy at three MFMAs.
//
// Backward kernels recompute the cheap intermediates from what the forward pass saved (x, rstd, the pre-norm sum m,
// the scan output) and write the operand pairs of every weight gradient for the batched weight-gradient launch.
#include <algorithm>

#include "hmx_common.h"
#include "profile.h"

#pragma clang fp contract(fast)

namespace {

using short8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_(float x) { return x * sigmoid_(x); }
__device__ __forceinline__ float silu_grad_(float x) {
    const float s = sigmoid_(x);
    return s * (1.f + x * (1.f - s));
}

struct Frag {
    short8 hi, lo;
};

// eight fp32 values -> bf16 hi / lo fragment (x = hi + lo to 16 significant bits)
__device__ __forceinline__ Frag make_frag(const float (&v)[8]) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hw[e] = hmx_pack_bf16x2(v[2 * e], v[2 * e + 1]);
        lw[e] = hmx_pack_bf16x2(v[2 * e] - __uint_as_float(hw[e] << 16), v[2 * e + 1] - __uint_as_float(hw[e] & 0xffff0000u));
    }
    Frag f;
    f.hi = __builtin_bit_cast(short8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
    f.lo = __builtin_bit_cast(short8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
    return f;
}

// Accumulator tile (32 features x 32 tokens) -> the B fragments of its two k-steps.  Register 8 s + j of lane half h is
// feature 16 s + 8 (j >> 2) + 4 h + (j & 3) of the tile: the k order the packed weights use.
__device__ __forceinline__ void tile_frags(const f32x16& a, Frag& f0, Frag& f1) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a[j];
    f0 = make_frag(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a[8 + j];
    f1 = make_frag(v);
}

// B fragment of k-step ks straight from a row-major fp32 row: columns 16 ks + 4 h + (0..3) and + 8.  Load and
// conversion are separate so that a loop can request several k-steps before it uses the first: with one or two waves
// per SIMD every load that is waited for alone costs a full memory round trip.
struct RawFrag {
    float4 a, b;
};
__device__ __forceinline__ RawFrag row_raw(const float* row, int ks, int h) {
    RawFrag r;
    r.a = *reinterpret_cast<const float4*>(row + 16 * ks + 4 * h);
    r.b = *reinterpret_cast<const float4*>(row + 16 * ks + 8 + 4 * h);
    return r;
}
__device__ __forceinline__ Frag raw_frag(const RawFrag& r) {
    const float v[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    return make_frag(v);
}
constexpr int kRowBatch = 4;   // k-steps requested together (KSe = ED / 16 is a multiple of 8)

// Where a kernel reads packed weights from: the pack buffer in global memory (L2), or -- for the entry range
// [lo, hi) that the workgroup copied at its start -- LDS.
extern __shared__ __attribute__((aligned(16))) uint4 w_lds[];
template <bool LDSW>
struct WSrc {
    const uint4* g;
    unsigned lo, hi;
};

// acc += W(tile nt, k-step ks) * B   with W packed as [(nt * KS + ks) * 2 + part][lane] 16-byte entries starting at
// entry `w` of the pack buffer
template <bool LDSW>
__device__ __forceinline__ void mma(f32x16& acc, const WSrc<LDSW>& src, size_t w, int nt, int KS, int ks, const Frag& b, int lane) {
    const unsigned base = (unsigned)w + (unsigned)(nt * KS + ks) * 128u;      // wave-uniform
    uint4 e0, e1;
    if (LDSW && base >= src.lo && base < src.hi) {
        const uint4* lp = w_lds + (base - src.lo);
        e0 = lp[lane];
        e1 = lp[64 + lane];
    } else {
        const uint4* wp = src.g + base;      // wave-uniform base (scalar registers) + one per-lane offset for every load
        e0 = wp[lane];
        e1 = wp[64 + lane];
    }
    const short8 a_hi = __builtin_bit_cast(short8, e0);
    const short8 a_lo = __builtin_bit_cast(short8, e1);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.hi, acc, 0, 0, 0);
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Tile nt of a row-major (.., ld) fp32 row in accumulator layout: register 4 q + e <-> column 32 nt + 8 q + 4 h + e
__device__ __forceinline__ void load_tile(f32x16& a, const float* row, int nt, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(row + 32 * nt + 8 * q + 4 * h);
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void store_tile(const f32x16& a, float* row, int nt, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(row + 32 * nt + 8 * q + 4 * h) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// Sequence position of token 32 * tile + i.  The 64-bit remainder is taken once per wave on the tile's first token (scalar
// unit); the lane's part is 32-bit.  Lanes past the last token get some valid position: their results are discarded.
__device__ __forceinline__ int tile_pos(int64_t tile, int i, int L) {
    const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)((tile * 32) % L));
    return (int)((first + (unsigned)i) % (unsigned)L);
}

// sum of a per-lane value over the two lanes (l, l ^ 32) that share a token
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32); }

// ---------------------------------------------------------------------------------------------------------------
// Packed weights of one block.  All offsets in 16-byte entries.  "T" = the transposed product of the backward pass.
struct PackLayout {
    int d, H, hd, ED, NT, NTh, KSd, KSh, KSe, KSz;
    // forward: W1, W2, Wg, Wu, Wd (d x d), Win[h] (2 ED x hd), Wout[h] (hd x ED)
    size_t w1, w2, wg, wu, wd, win, wout, win_stride, wout_stride;
    // transposed: W1T .. WdT (d x d), WinT[h] (hd x 2 ED), WoutT[h] (ED x hd)
    size_t w1t, w2t, wgt, wut, wdt, wint, woutt, wint_stride, woutt_stride;
    size_t vec;         // four d-vectors of floats (scale1, b1, b2, scale2) in the widened layout
    size_t total;
};

__host__ __device__ inline size_t pack_entries(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 2 * 64; }

__host__ __device__ inline PackLayout make_layout(int d, int H) {
    PackLayout l{};
    l.d = d; l.H = H; l.hd = d / H; l.ED = 4 * l.hd;
    l.NT = d / 32; l.NTh = l.hd / 32; l.KSd = d / 16; l.KSh = l.hd / 16; l.KSe = l.ED / 16; l.KSz = 2 * l.ED / 16;
    size_t o = 0;
    const size_t dd = pack_entries(d, d);
    // Order: what one kernel reads most is contiguous, so that it can sit in LDS as one range:
    //   pre forward [win .. w1], pre backward [w1 .. wint], post forward [wout .. wu], post backward [wg .. w2t]
    l.win_stride = pack_entries(2 * l.ED, l.hd); l.win = o; o += l.win_stride * H;
    l.w1 = o; o += dd; l.w1t = o; o += dd;
    l.wint_stride = pack_entries(l.hd, 2 * l.ED); l.wint = o; o += l.wint_stride * H;
    l.wout_stride = pack_entries(l.hd, l.ED); l.wout = o; o += l.wout_stride * H;
    l.w2 = o; o += dd; l.wd = o; o += dd; l.wg = o; o += dd; l.wu = o; o += dd;
    l.wdt = o; o += dd; l.wgt = o; o += dd; l.wut = o; o += dd; l.w2t = o; o += dd;
    l.woutt_stride = pack_entries(l.ED, l.hd); l.woutt = o; o += l.woutt_stride * H;
    l.vec = o; o += (size_t)d;          // 4 d floats = d entries of 16 bytes
    l.total = o;
    return l;
}

// A weight whose rows / columns come in groups that the packed image widens with zeros (head_dim 30 -> 32 and so on):
// row n of the image is row (n / gNp) * gN + n % gNp of W when n % gNp < gN, a zero row otherwise; same for columns.
struct PackJob {
    const float* W;     // null: vector job (K = 1, plain floats at entry `out`)
    int64_t rs, cs;     // Wv(n, k) = W[n * rs + k * cs]
    int N, K;           // image size
    int gN, gNp, gK, gKp;
    size_t out;         // entry offset
};
constexpr int kMaxPackJobs = 2 * (5 + 2 * 8) + 4;
struct PackParams {
    PackJob job[kMaxPackJobs];
    int n_jobs;
    uint4* out;
};

// entry (nt, ks, part, lane): element j = Wv(32 nt + (lane & 31), 16 ks + 8 (j >> 2) + 4 (lane >> 5) + (j & 3))
__global__ __launch_bounds__(256) void block_pack_kernel(PackParams p) {
    const PackJob& jb = p.job[blockIdx.y];
    if (jb.gK == 0) {           // vector: image element n <- W[group-mapped n], plain fp32
        float* dst = reinterpret_cast<float*>(p.out + jb.out);
        for (int n = blockIdx.x * 256 + threadIdx.x; n < jb.N; n += gridDim.x * 256) {
            const int r = n % jb.gNp;
            dst[n] = r < jb.gN ? jb.W[(n / jb.gNp) * jb.gN + r] : 0.f;
        }
        return;
    }
    const int KS = (jb.K + 15) / 16, NTo = (jb.N + 31) / 32;
    const int n_frag = NTo * KS * 64;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n_frag; idx += gridDim.x * 256) {
        const int lane = idx & 63, ks = (idx >> 6) % KS, nt = (idx >> 6) / KS;
        const int n = 32 * nt + (lane & 31), h = lane >> 5;
        const int nr = n % jb.gNp;
        const bool n_ok = n < jb.N && nr < jb.gN;
        const int64_t n_src = (int64_t)(n / jb.gNp) * jb.gN + nr;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
            const int kr = k % jb.gKp;
            const int64_t k_src = (int64_t)(k / jb.gKp) * jb.gK + kr;
            v[j] = (n_ok && k < jb.K && kr < jb.gK) ? jb.W[n_src * jb.rs + k_src * jb.cs] : 0.f;
        }
        const Frag f = make_frag(v);
        uint4* e = p.out + jb.out + ((size_t)(nt * KS + ks) * 2) * 64 + lane;
        e[0] = __builtin_bit_cast(uint4, f.hi);
        e[64] = __builtin_bit_cast(uint4, f.lo);
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct BlockParams {
    int64_t T;
    int L, first;
    float eps, inv_sqrt_h;
    float inv_d;            // 1 / (true model width): the RMSNorm means run over the true features of a widened layout
    PackLayout lay;
    const uint4* pk;
    unsigned lds_lo, lds_hi;  // entry range of the pack buffer a workgroup keeps in LDS (template flag LDSW)
    const float* mask;      // (T)
    // pre half
    const float* x;         // (T, d) block input
    const float* drop;      // (T, d) dropout multiplier or null
    const float* scale1;    // (d)
    const float* b1;        // (d)
    const float* cs;        // rope cos (L, hd / 2)
    const float* sn;        // rope sin
    float* rstd1;           // (T)
    float* x_in;            // (T, C)
    float* z;               // (T, C)
    float* y2;              // (T, d)
    // post half
    const float* yg;        // (T, C)
    const float* b2;        // (d)
    const float* scale2;    // (d)
    float* out;             // (T, d)
    float* m;               // (T, d)  x_org + W2 p + b2 + residual (input of RMSNorm2)
    float* rstd2;           // (T)
    // backward, post half
    const float* dout;      // (T, d)
    float* dyg;             // (T, C)
    float* dy2;             // (T, d)
    float* dm;              // (T, d)
    float* hmid;            // (T, d)   wgrad operands
    float* n2;
    float* dg;
    float* du;
    float* p;
    float* dy1;
    float* dscale_part;     // (scale_rows, d): one row per wave of the launch, the rows past them zeroed by the launch
    int scale_rows;
    // backward, pre half
    const float* dx_in;     // (T, C)
    const float* dz;        // (T, C)
    float* dx;              // (T, d)
    float* s;               // (T, d)   wgrad operands
    float* dx1;
    float* n;
};

// The partial-sum buffer has a row for every wave of the largest launch; a smaller launch clears the rows its waves do
// not write, so the caller sums all rows of an uninitialised buffer.
__device__ __forceinline__ void zero_unwritten_scale_rows(const BlockParams& p, int waves_per_block) {
    const int64_t first = (int64_t)gridDim.x * waves_per_block * p.lay.d, end = (int64_t)p.scale_rows * p.lay.d;
    for (int64_t e = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < end; e += (int64_t)gridDim.x * blockDim.x)
        p.dscale_part[e] = 0.f;
}

constexpr int kMaxNT = 4;   // d <= 128

// dropout multiplier, RMSNorm1: fills xd (x * drop), returns rstd; n = xd * rstd * scale1
template <int NT>
__device__ __forceinline__ float norm1(const BlockParams& p, const float* xrow, const float* droprow, int h,
                                       f32x16 (&xd)[NT], f32x16 (&n)[NT]) {
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        load_tile(xd[nt], xrow, nt, h);
        if (droprow != nullptr) {
            f32x16 dr;
            load_tile(dr, droprow, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) xd[nt][r] *= dr[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += xd[nt][r] * xd[nt][r];
    }
    ss = pair_sum(ss);
    const float rstd = 1.f / sqrtf(ss * p.inv_d + p.eps);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x16 sc;
        load_tile(sc, p.scale1, nt, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) n[nt][r] = xd[nt][r] * rstd * sc[r];
    }
    return rstd;
}

// x1 = W1 (n * mask) + b1
template <int NT, bool LDSW>
__device__ __forceinline__ void linear1(const BlockParams& p, const WSrc<LDSW>& src, const f32x16 (&n)[NT], float mk, int h, int lane,
                                        f32x16 (&x1)[NT]) {
    Frag nf[2 * NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x16 t = n[nt];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] *= mk;
        tile_frags(t, nf[2 * nt], nf[2 * nt + 1]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        load_tile(x1[nt], p.b1, nt, h);
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) mma<LDSW>(x1[nt], src, p.lay.w1, nt, 2 * NT, ks, nf[ks], lane);
    }
}

// ------------------------------------------------------------------ pre half, forward
template <int NT, bool LDSW>
__global__ __launch_bounds__(256) void block_pre_fwd_kernel(BlockParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const PackLayout& l = p.lay;
    const WSrc<LDSW> src{p.pk, p.lds_lo, p.lds_hi};
    if (LDSW) {
        for (unsigned idx = threadIdx.x; idx < p.lds_hi - p.lds_lo; idx += 256) w_lds[idx] = p.pk[p.lds_lo + idx];
        __syncthreads();
    }
    const int d = 32 * NT, C = l.H * l.ED;
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        f32x16 xd[NT], n[NT], x1[NT];
        const float rstd = norm1<NT>(p, p.x + tt * d, p.drop ? p.drop + tt * d : nullptr, h, xd, n);
        linear1<NT, LDSW>(p, src, n, p.mask ? p.mask[tt] : 1.f, h, lane, x1);
        // s = SiLU(x1) (kept in x1), y2 = RoPE(s): pairs (2 i, 2 i + 1) are registers (e, e + 1), e even
        const int pos = tile_pos(tile, i, p.L);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 y2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * nt + 8 * q + 4 * h;                   // first feature of this register quad
                const int ip = (f0 % l.hd) / 2;
                const float2 c = *reinterpret_cast<const float2*>(p.cs + pos * (l.hd / 2) + ip);
                const float2 sv = *reinterpret_cast<const float2*>(p.sn + pos * (l.hd / 2) + ip);
                const float s0 = silu_(x1[nt][4 * q]), s1 = silu_(x1[nt][4 * q + 1]);
                const float s2 = silu_(x1[nt][4 * q + 2]), s3 = silu_(x1[nt][4 * q + 3]);
                x1[nt][4 * q] = s0; x1[nt][4 * q + 1] = s1; x1[nt][4 * q + 2] = s2; x1[nt][4 * q + 3] = s3;
                y2[4 * q] = s0 * c.x - s1 * sv.x;
                y2[4 * q + 1] = s0 * sv.x + s1 * c.x;
                y2[4 * q + 2] = s2 * c.y - s3 * sv.y;
                y2[4 * q + 3] = s2 * sv.y + s3 * c.y;
            }
            if (ok) store_tile(y2, p.y2 + t * d, nt, h);
        }
        // in_proj per head: [x_in | z] = Win_h s_h, streamed one 32-channel output tile at a time
        Frag sf[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tile_frags(x1[nt], sf[2 * nt], sf[2 * nt + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int hh = 0; hh < l.H; ++hh) {
            const size_t w = l.win + hh * l.win_stride;
#pragma unroll 1
            for (int ot = 0; ot < 2 * l.ED / 32; ++ot) {
                f32x16 acc;
                zero(acc);
#pragma unroll
                for (int ks = 0; ks < 2 * NT; ++ks)
                    if (ks >= hh * l.KSh && ks < (hh + 1) * l.KSh) mma<LDSW>(acc, src, w, ot, l.KSh, ks - hh * l.KSh, sf[ks], lane);
                if (ok) {
                    const int ch = ot * 32;
                    float* dst = (ch < l.ED ? p.x_in : p.z) + t * C + hh * l.ED + (ch < l.ED ? ch : ch - l.ED);
                    store_tile(acc, dst, 0, h);
                }
            }
        }
        if (ok && h == 0) p.rstd1[t] = rstd;
    }
}

// ------------------------------------------------------------------ post half, forward
// y1 = Wout_h yg_h for every head (feature tiles of head h: [h * NTh, (h + 1) * NTh))
template <int NT, bool LDSW>
__device__ __forceinline__ void out_proj(const BlockParams& p, const WSrc<LDSW>& src, const float* ygrow, int h, int lane,
                                         f32x16 (&y1)[NT]) {
    const PackLayout& l = p.lay;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) zero(y1[nt]);
#pragma unroll 1
    for (int hh = 0; hh < l.H; ++hh) {
        const size_t w = l.wout + hh * l.wout_stride;

        for (int ks0 = 0; ks0 < l.KSe; ks0 += kRowBatch) {
            RawFrag raw[kRowBatch];
#pragma unroll
            for (int u = 0; u < kRowBatch; ++u) raw[u] = row_raw(ygrow + hh * l.ED, ks0 + u, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kRowBatch; ++u) {
                const Frag f = raw_frag(raw[u]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (nt >= hh * l.NTh && nt < (hh + 1) * l.NTh)
                        mma<LDSW>(y1[nt], src, w, nt - hh * l.NTh, l.KSe, ks0 + u, f, lane);
            }
        }
    }
}

template <int NT, bool LDSW>
__global__ __launch_bounds__(256, NT <= 2 ? 2 : 1) void block_post_fwd_kernel(BlockParams p) {   // d <= 64: two workgroups per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const PackLayout& l = p.lay;
    const WSrc<LDSW> src{p.pk, p.lds_lo, p.lds_hi};
    if (LDSW) {
        for (unsigned idx = threadIdx.x; idx < p.lds_hi - p.lds_lo; idx += 256) w_lds[idx] = p.pk[p.lds_lo + idx];
        __syncthreads();
    }
    const int d = 32 * NT, C = l.H * l.ED;
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        f32x16 m[NT];
        {
            f32x16 y1[NT];
            out_proj<NT, LDSW>(p, src, p.yg + tt * C, h, lane, y1);
            Frag pf[2 * NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {      // p = y1 * y2 / sqrt(H)
                f32x16 y2;
                load_tile(y2, p.y2 + tt * d, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) y1[nt][r] *= y2[r] * p.inv_sqrt_h;
                tile_frags(y1[nt], pf[2 * nt], pf[2 * nt + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // m = W2 p + b2 + x_org + (first ? n : x)
            const float rstd1 = p.first ? p.rstd1[tt] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                load_tile(m[nt], p.b2, nt, h);
#pragma unroll
                for (int ks = 0; ks < 2 * NT; ++ks) mma<LDSW>(m[nt], src, l.w2, nt, 2 * NT, ks, pf[ks], lane);
                f32x16 xv;
                load_tile(xv, p.x + tt * d, nt, h);
                if (p.first) {
                    f32x16 sc;
                    load_tile(sc, p.scale1, nt, h);
                    if (p.drop != nullptr) {
                        f32x16 dr;
                        load_tile(dr, p.drop + tt * d, nt, h);
#pragma unroll
                        for (int r = 0; r < 16; ++r) m[nt][r] += xv[r] + xv[r] * dr[r] * rstd1 * sc[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) m[nt][r] += xv[r] + xv[r] * rstd1 * sc[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[nt][r] += 2.f * xv[r];
                }
                if (ok) store_tile(m[nt], p.m + t * d, nt, h);
            }
        }
        // RMSNorm2
        float ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ss += m[nt][r] * m[nt][r];
        ss = pair_sum(ss);
        const float rstd2 = 1.f / sqrtf(ss * p.inv_d + p.eps);
        f32x16 n2[NT];
        Frag nf[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 sc;
            load_tile(sc, p.scale2, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) n2[nt][r] = m[nt][r] * rstd2 * sc[r];
            tile_frags(n2[nt], nf[2 * nt], nf[2 * nt + 1]);
        }
        // out = Wd (SiLU(Wg n2) * (Wu n2)) + (first ? n2 : m), streamed over the hidden tiles
        f32x16 o[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = p.first ? n2[nt] : m[nt];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int ht = 0; ht < NT; ++ht) {       // a real loop: bounds how far the compiler hoists the weight loads
            f32x16 g, u;
            zero(g);
            zero(u);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) {
                mma<LDSW>(g, src, l.wg, ht, 2 * NT, ks, nf[ks], lane);
                mma<LDSW>(u, src, l.wu, ht, 2 * NT, ks, nf[ks], lane);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = silu_(g[r]) * u[r];
            Frag h0, h1;
            tile_frags(g, h0, h1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mma<LDSW>(o[nt], src, l.wd, nt, 2 * NT, 2 * ht, h0, lane);
                mma<LDSW>(o[nt], src, l.wd, nt, 2 * NT, 2 * ht + 1, h1, lane);
            }
        }
        if (ok) {
            const float mk = p.mask ? p.mask[t] : 1.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[nt][r] *= mk;
                store_tile(o[nt], p.out + t * d, nt, h);
            }
            if (h == 0) p.rstd2[t] = rstd2;
        }
    }
}

// Column sums over a wave's tokens of per-lane partials held in accumulator layout -> one row of (waves, d).
template <int NT>
__device__ __forceinline__ void write_scale_partial(const f32x16 (&acc)[NT], float* part_row, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[nt][r];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);     // over the 32 token lanes of a half
            if ((lane & 31) == 0) part_row[32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h] = v;
        }
}

// Per-lane running sums of a backward half's scale gradient, one value per accumulator register of NT tiles.  In
// registers up to d = 96; at d = 128 the 64 values per lane live in LDS instead (free there: the weights of that width
// do not fit LDS and are read from L2), laid out [value][thread] so that a wave's accesses are conflict-free.
template <int NT, bool IN_LDS>
struct ColumnSums {
    f32x16 reg[IN_LDS ? 1 : NT];
    float* mine;
    static constexpr int stride = 256;          // threads of a backward workgroup
    __device__ __forceinline__ void init() {
        if constexpr (IN_LDS) {
            mine = reinterpret_cast<float*>(w_lds) + threadIdx.x;
#pragma unroll
            for (int v = 0; v < 16 * NT; ++v) mine[v * stride] = 0.f;
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) zero(reg[nt]);
        }
    }
    __device__ __forceinline__ void add(int nt, const f32x16& v) {
        if constexpr (IN_LDS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(16 * nt + r) * stride] += v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) reg[nt][r] += v[r];
        }
    }
    __device__ __forceinline__ void flush(float* part_row, int lane) {
        if constexpr (IN_LDS) {
            f32x16 all[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) all[nt][r] = mine[(16 * nt + r) * stride];
            write_scale_partial<NT>(all, part_row, lane);
        } else {
            write_scale_partial<NT>(reg, part_row, lane);
        }
    }
};
constexpr bool sums_in_lds(int NT, bool LDSW) { return NT >= 4 && !LDSW; }
constexpr size_t kSumsLdsBytes = 256 * 64 * sizeof(float);      // 256 threads x 64 values

// Waves per workgroup of the backward halves.  Their working set wants more than 256 registers, i.e. one wave per SIMD.
// Eight waves per workgroup (two per SIMD at 256 registers each, sharing one LDS copy of the weights) were measured
// at d = 64: 216 / 316 bytes per lane spill to scratch and both halves get slower (post 172 -> 177 us, pre 123 -> 139 us
// per launch on average), so four it stays.
__host__ __device__ constexpr int bwd_waves(int /*NT*/) { return 4; }

// ------------------------------------------------------------------ post half, backward
template <int NT, bool LDSW>
__global__ __launch_bounds__(64 * bwd_waves(NT)) void block_post_bwd_kernel(BlockParams p) {
    constexpr int WPB = bwd_waves(NT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const PackLayout& l = p.lay;
    const WSrc<LDSW> src{p.pk, p.lds_lo, p.lds_hi};
    if (LDSW) {
        for (unsigned idx = threadIdx.x; idx < p.lds_hi - p.lds_lo; idx += 64 * WPB) w_lds[idx] = p.pk[p.lds_lo + idx];
        __syncthreads();
    }
    const int d = 32 * NT, C = l.H * l.ED;
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    ColumnSums<NT, sums_in_lds(NT, LDSW)> dsc;
    dsc.init();
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        const float mk = (p.mask ? p.mask[tt] : 1.f) * (ok ? 1.f : 0.f);
        const float rstd2 = p.rstd2[tt];
        // Only fragments stay live across a phase; m, scale2 and dout are re-read (L1 / L2 hits) where a later phase
        // needs them again: the register file holds the fragments, the accumulators and one phase's temporaries.
        Frag nf[2 * NT], dof[2 * NT];
        f32x16 dn2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 mt, sc, n2, dot_;
            load_tile(mt, p.m + tt * d, nt, h);
            load_tile(sc, p.scale2, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) n2[r] = mt[r] * rstd2 * sc[r];
            tile_frags(n2, nf[2 * nt], nf[2 * nt + 1]);
            if (ok) store_tile(n2, p.n2 + t * d, nt, h);
            load_tile(dot_, p.dout + tt * d, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) dot_[r] *= mk;                     // do = dout * mask
            tile_frags(dot_, dof[2 * nt], dof[2 * nt + 1]);
            if (p.first) dn2[nt] = dot_;                                   // out = f + n2
            else zero(dn2[nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // FFN backward, streamed over the hidden tiles: recompute g, u; dhmid = Wd^T do; dn2 += Wg^T dg + Wu^T du
#pragma unroll 1
        for (int ht = 0; ht < NT; ++ht) {
            f32x16 g, u, dh;
            zero(g);
            zero(u);
            zero(dh);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) {
                mma<LDSW>(g, src, l.wg, ht, 2 * NT, ks, nf[ks], lane);
                mma<LDSW>(u, src, l.wu, ht, 2 * NT, ks, nf[ks], lane);
                mma<LDSW>(dh, src, l.wdt, ht, 2 * NT, ks, dof[ks], lane);
            }
            f32x16 hm, dg, du;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sg = silu_(g[r]);
                hm[r] = sg * u[r];
                dg[r] = dh[r] * u[r] * silu_grad_(g[r]);
                du[r] = dh[r] * sg;
            }
            if (ok) {
                store_tile(hm, p.hmid + t * d, ht, h);
                store_tile(dg, p.dg + t * d, ht, h);
                store_tile(du, p.du + t * d, ht, h);
            }
            Frag g0, g1, u0, u1;
            tile_frags(dg, g0, g1);
            tile_frags(du, u0, u1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mma<LDSW>(dn2[nt], src, l.wgt, nt, 2 * NT, 2 * ht, g0, lane);
                mma<LDSW>(dn2[nt], src, l.wgt, nt, 2 * NT, 2 * ht + 1, g1, lane);
                mma<LDSW>(dn2[nt], src, l.wut, nt, 2 * NT, 2 * ht, u0, lane);
                mma<LDSW>(dn2[nt], src, l.wut, nt, 2 * NT, 2 * ht + 1, u1, lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // RMSNorm2 backward: dm = rstd (s dn2 - m rstd^2 mean(dn2 s m)) (+ do in a regular block: out = f + m)
        float dot = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 mt, sc;
            load_tile(mt, p.m + tt * d, nt, h);
            load_tile(sc, p.scale2, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += dn2[nt][r] * sc[r] * mt[r];
        }
        dot = pair_sum(dot);
        const float coef = dot * rstd2 * rstd2 * p.inv_d;
        Frag dmf[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 mt, sc, dm, dsv;
            load_tile(mt, p.m + tt * d, nt, h);
            load_tile(sc, p.scale2, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dsv[r] = dn2[nt][r] * mt[r] * rstd2;                        // masked / padded tokens contribute dn2 = 0
                dm[r] = rstd2 * (sc[r] * dn2[nt][r] - mt[r] * coef);
            }
            dsc.add(nt, dsv);
            if (!p.first) {
                f32x16 dot_;
                load_tile(dot_, p.dout + tt * d, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) dm[r] += dot_[r] * mk;
            }
            if (ok) store_tile(dm, p.dm + t * d, nt, h);
            tile_frags(dm, dmf[2 * nt], dmf[2 * nt + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // mixer tail backward: dp = W2^T dm; recompute y1; dy1 = dp y2 / sqrt(H), dy2 = dp y1 / sqrt(H); dyg = Wout^T dy1
        f32x16 y1[NT];
        out_proj<NT, LDSW>(p, src, p.yg + tt * C, h, lane, y1);
        Frag d1f[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 dp, y2;
            zero(dp);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) mma<LDSW>(dp, src, l.w2t, nt, 2 * NT, ks, dmf[ks], lane);
            load_tile(y2, p.y2 + tt * d, nt, h);
            f32x16 dy1, dy2, pp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pp[r] = y1[nt][r] * y2[r] * p.inv_sqrt_h;
                dy1[r] = dp[r] * y2[r] * p.inv_sqrt_h;
                dy2[r] = dp[r] * y1[nt][r] * p.inv_sqrt_h;
            }
            if (ok) {
                store_tile(pp, p.p + t * d, nt, h);
                store_tile(dy1, p.dy1 + t * d, nt, h);
                store_tile(dy2, p.dy2 + t * d, nt, h);
            }
            tile_frags(dy1, d1f[2 * nt], d1f[2 * nt + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int hh = 0; hh < l.H; ++hh) {
            const size_t w = l.woutt + hh * l.woutt_stride;
#pragma unroll 1
            for (int ot = 0; ot < l.ED / 32; ++ot) {
                f32x16 acc;
                zero(acc);
#pragma unroll
                for (int ks = 0; ks < 2 * NT; ++ks)
                    if (ks >= hh * l.KSh && ks < (hh + 1) * l.KSh) mma<LDSW>(acc, src, w, ot, l.KSh, ks - hh * l.KSh, d1f[ks], lane);
                if (ok) store_tile(acc, p.dyg + t * C + hh * l.ED + ot * 32, 0, h);
            }
        }
    }
    dsc.flush(p.dscale_part + ((int64_t)blockIdx.x * WPB + wave) * d, lane);
    zero_unwritten_scale_rows(p, WPB);
}

// ------------------------------------------------------------------ pre half, backward
template <int NT, bool LDSW>
__global__ __launch_bounds__(64 * bwd_waves(NT)) void block_pre_bwd_kernel(BlockParams p) {
    constexpr int WPB = bwd_waves(NT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const PackLayout& l = p.lay;
    const WSrc<LDSW> src{p.pk, p.lds_lo, p.lds_hi};
    if (LDSW) {
        for (unsigned idx = threadIdx.x; idx < p.lds_hi - p.lds_lo; idx += 64 * WPB) w_lds[idx] = p.pk[p.lds_lo + idx];
        __syncthreads();
    }
    const int d = 32 * NT, C = l.H * l.ED;
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    ColumnSums<NT, sums_in_lds(NT, LDSW)> dsc;
    dsc.init();
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        const float live = ok ? 1.f : 0.f;
        const float mk = p.mask ? p.mask[tt] : 1.f;
        // Phases in the order that keeps the fewest tiles alive: the in_proj gradient first (its accumulators are all
        // that lives through the long streaming loop), then the recomputation of n and x1, one x1 tile at a time.
        // ds = sum_heads Win_h^T [dx_in_h ; dz_h]
        f32x16 ds[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) zero(ds[nt]);
#pragma unroll 1
        for (int hh = 0; hh < l.H; ++hh) {
            const size_t w = l.wint + hh * l.wint_stride;

            for (int ks0 = 0; ks0 < l.KSz; ks0 += kRowBatch) {      // a batch lies in dx_in or in dz, never across
                const float* grow = (ks0 < l.KSe ? p.dx_in : p.dz) + tt * C + hh * l.ED;
                const int kr = ks0 < l.KSe ? ks0 : ks0 - l.KSe;
                RawFrag raw[kRowBatch];
#pragma unroll
                for (int u = 0; u < kRowBatch; ++u) raw[u] = row_raw(grow, kr + u, h);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < kRowBatch; ++u) {
                    const Frag f = raw_frag(raw[u]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (nt >= hh * l.NTh && nt < (hh + 1) * l.NTh)
                            mma<LDSW>(ds[nt], src, w, nt - hh * l.NTh, l.KSz, ks0 + u, f, lane);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // n = RMSNorm1(x * drop): written for the weight gradient of linear1 and kept as the fragments of n * mask.
        // The tiles of x * drop stay in registers where the register file has room for them (d <= 64); wider blocks
        // read x (and drop) again in the two places of the norm's backward that use them -- L1 / L2 hits.
        constexpr bool kKeepXd = NT <= 2;
        const float* xrow = p.x + tt * d;
        const float* droprow = p.drop ? p.drop + tt * d : nullptr;
        f32x16 xd[NT];
        Frag nf[2 * NT];
        float rstd;
        {
            f32x16 n[NT];
            rstd = norm1<NT>(p, xrow, droprow, h, xd, n);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (ok) store_tile(n[nt], p.n + t * d, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) n[nt][r] *= mk;
                tile_frags(n[nt], nf[2 * nt], nf[2 * nt + 1]);
            }
        }
        auto xd_tile = [&](int nt) -> f32x16 {
            if constexpr (kKeepXd) return xd[nt];
            f32x16 v;
            load_tile(v, xrow, nt, h);
            if (droprow != nullptr) {
                f32x16 dr;
                load_tile(dr, droprow, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= dr[r];
            }
            return v;
        };
        if constexpr (!kKeepXd) __builtin_amdgcn_sched_barrier(0);
        // x1 = W1 (n * mask) + b1 one tile at a time;  + RoPE^T(dy2);  dx1 = ds * SiLU'(x1);  s = SiLU(x1)
        const int pos = tile_pos(tile, i, p.L);
        Frag d1f[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 x1, g2, sv_, dx1;
            load_tile(x1, p.b1, nt, h);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) mma<LDSW>(x1, src, l.w1, nt, 2 * NT, ks, nf[ks], lane);
            load_tile(g2, p.dy2 + tt * d, nt, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * nt + 8 * q + 4 * h;
                const int ip = (f0 % l.hd) / 2;
                const float2 c = *reinterpret_cast<const float2*>(p.cs + pos * (l.hd / 2) + ip);
                const float2 sn = *reinterpret_cast<const float2*>(p.sn + pos * (l.hd / 2) + ip);
                const float ge0 = ds[nt][4 * q] + g2[4 * q] * c.x + g2[4 * q + 1] * sn.x;
                const float go0 = ds[nt][4 * q + 1] - g2[4 * q] * sn.x + g2[4 * q + 1] * c.x;
                const float ge1 = ds[nt][4 * q + 2] + g2[4 * q + 2] * c.y + g2[4 * q + 3] * sn.y;
                const float go1 = ds[nt][4 * q + 3] - g2[4 * q + 2] * sn.y + g2[4 * q + 3] * c.y;
                dx1[4 * q] = ge0 * silu_grad_(x1[4 * q]) * live;
                dx1[4 * q + 1] = go0 * silu_grad_(x1[4 * q + 1]) * live;
                dx1[4 * q + 2] = ge1 * silu_grad_(x1[4 * q + 2]) * live;
                dx1[4 * q + 3] = go1 * silu_grad_(x1[4 * q + 3]) * live;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sv_[r] = silu_(x1[r]);
            if (ok) {
                store_tile(sv_, p.s + t * d, nt, h);
                store_tile(dx1, p.dx1 + t * d, nt, h);
            }
            tile_frags(dx1, d1f[2 * nt], d1f[2 * nt + 1]);
            if constexpr (!kKeepXd) __builtin_amdgcn_sched_barrier(0);      // one tile's weight loads in flight at a time
        }
        // dn = (W1^T dx1) * mask (+ dm in the first block: m = n + ...);  RMSNorm1 backward;  dx = dxd * drop + c dm
        f32x16 dn[NT];
        float dot = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            zero(dn[nt]);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) mma<LDSW>(dn[nt], src, l.w1t, nt, 2 * NT, ks, d1f[ks], lane);
            f32x16 sc1;
            load_tile(sc1, p.scale1, nt, h);
            const f32x16 xv = xd_tile(nt);
            if (p.first) {
                f32x16 dmv;
                load_tile(dmv, p.dm + tt * d, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) dn[nt][r] = dn[nt][r] * mk + dmv[r] * live;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) dn[nt][r] *= mk;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += dn[nt][r] * sc1[r] * xv[r];
            if constexpr (!kKeepXd) __builtin_amdgcn_sched_barrier(0);
        }
        dot = pair_sum(dot);
        const float coef = dot * rstd * rstd * p.inv_d;
        const float cres = p.first ? 1.f : 2.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 dxv, sc1, dmv, dsv;
            load_tile(sc1, p.scale1, nt, h);
            load_tile(dmv, p.dm + tt * d, nt, h);
            const f32x16 xv = xd_tile(nt);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dsv[r] = dn[nt][r] * xv[r] * rstd;
                dxv[r] = rstd * (sc1[r] * dn[nt][r] - xv[r] * coef);
            }
            dsc.add(nt, dsv);
            if (droprow != nullptr) {
                f32x16 dr;
                load_tile(dr, droprow, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) dxv[r] *= dr[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dxv[r] += cres * dmv[r] * live;
            if (ok) store_tile(dxv, p.dx + t * d, nt, h);
        }
    }
    dsc.flush(p.dscale_part + ((int64_t)blockIdx.x * WPB + wave) * d, lane);
    zero_unwritten_scale_rows(p, WPB);
}

// ------------------------------------------------------------------ input assembly (hydra3.py:621-669)
// seqs[t] = ((Wh[:, :ds] (table_row(t) + pos_row(t)) + bh + pre[t]) * sqrt(d) + time[tb(t)] + action[a(t)] + dwell[w(t)]) * pm(t)
// One wave per 32 tokens, tokens on the MFMA columns like the block halves.  The workgroup converts the projection
// weight to bf16 hi / lo fragments in LDS at its start (16 KB at 64 x 64); a lane then reads ITS token's two table rows
// k-step by k-step straight into B fragments, so the gathered rows never exist in memory (training keeps their sum
// for the weight gradient).  Index work of the reference (padding mask, positions, time-interval bins) is done on the way.
struct AssemblyParams {
    int64_t T;
    int L, ds, d, time_span;
    const int64_t *ids, *token_type, *ts, *action, *dwell;
    const float *item_table, *user_table, *pos_table;
    int64_t item_rows, user_rows, pos_rows;
    const float* W;          // (d, ldw), first ds columns
    int64_t ldw;
    const float* bias;       // (d)
    const float* pre;        // (T, d) or null
    float scale;
    const float *time_rows, *action_rows, *dwell_rows;   // (n, d) each
    int64_t time_n, action_n, dwell_n;
    float* out;              // (T, d)
    float* mask;             // (T)
    int64_t* pos_idx;        // (T)
    int64_t* time_bin;       // (T)
    float* e;                // (T, ds) or null: table row + position row
    // backward
    const float* dout;       // (T, d)
    const float* pmask;      // (T)
    float* g;                // (T, d)   dout * pm
    float* gs;               // (T, d)   g * scale
    float* de;               // (T, ds)  Wh[:, :ds]^T gs
};

// fragments of the (N x K) weight view Wv(n, k) = W[n * rs + k * cs] into w_lds, entry order of block_pack_kernel
__device__ __forceinline__ void weight_frags_to_lds(const float* W, int64_t rs, int64_t cs, int NTo, int KS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    for (int idx = wave; idx < NTo * KS; idx += blockDim.x >> 6) {
        const int nt = idx / KS, ks = idx % KS;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)(32 * nt + i) * rs + (int64_t)(16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)) * cs];
        const Frag f = make_frag(v);
        w_lds[(size_t)idx * 128 + lane] = __builtin_bit_cast(uint4, f.hi);
        w_lds[(size_t)idx * 128 + 64 + lane] = __builtin_bit_cast(uint4, f.lo);
    }
    __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(256) void input_assembly_fwd_kernel(AssemblyParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int KS = p.ds / 16, d = 32 * NT;
    weight_frags_to_lds(p.W, p.ldw, 1, NT, KS);
    const WSrc<true> src{reinterpret_cast<const uint4*>(p.W), 0u, (unsigned)(NT * KS * 128)};
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        // index preparation (hydra3.py:621, 629-635, 650-659)
        const int l = tile_pos(tile, i, p.L);
        const int64_t id = p.ids[tt];
        const bool valid = id != 0;
        const int64_t pos = valid ? l + 1 : 0;
        int64_t dt = 0;
        if (ok && l > 0 && p.ids[tt - 1] != 0) dt = p.ts[tt] - p.ts[tt - 1];
        int64_t bin = (int64_t)logf((float)dt + 1.0f);
        bin = bin > p.time_span - 1 ? p.time_span - 1 : bin;
        const int64_t tb = valid ? bin : 0;
        const float pm = valid ? 1.f : 0.f;
        const int64_t typ = p.token_type[tt];
        const float* row = nullptr;                     // embedding.py:436-450: item rows, user rows, zeros elsewhere
        if (typ == 1 && id >= 0 && id < p.item_rows) row = p.item_table + id * p.ds;
        else if (typ == 2 && id >= 0 && id < p.user_rows) row = p.user_table + id * p.ds;
        const float* prow = pos < p.pos_rows ? p.pos_table + pos * p.ds : nullptr;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_tile(acc[nt], p.bias, nt, h);
        for (int ks0 = 0; ks0 < KS; ks0 += 4) {          // ds is a multiple of 32: two or four k-steps requested together
            RawFrag ra[4], rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                ra[u].a = ra[u].b = rb[u].a = rb[u].b = z;
                if (ks0 + u < KS) {
                    if (row != nullptr) ra[u] = row_raw(row, ks0 + u, h);
                    if (prow != nullptr) rb[u] = row_raw(prow, ks0 + u, h);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks0 + u < KS) {
                    RawFrag r;
                    r.a = make_float4(ra[u].a.x + rb[u].a.x, ra[u].a.y + rb[u].a.y, ra[u].a.z + rb[u].a.z, ra[u].a.w + rb[u].a.w);
                    r.b = make_float4(ra[u].b.x + rb[u].b.x, ra[u].b.y + rb[u].b.y, ra[u].b.z + rb[u].b.z, ra[u].b.w + rb[u].b.w);
                    if (p.e != nullptr && ok) {
                        *reinterpret_cast<float4*>(p.e + t * p.ds + 16 * (ks0 + u) + 4 * h) = r.a;
                        *reinterpret_cast<float4*>(p.e + t * p.ds + 16 * (ks0 + u) + 8 + 4 * h) = r.b;
                    }
                    const Frag f = raw_frag(r);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma<true>(acc[nt], src, 0, nt, KS, ks0 + u, f, lane);
                }
            }
        }
        const int64_t a = p.action[tt], w = p.dwell[tt];
        const float* trow = tb >= 0 && tb < p.time_n ? p.time_rows + tb * d : nullptr;
        const float* arow = a >= 0 && a < p.action_n ? p.action_rows + a * d : nullptr;
        const float* wrow = w >= 0 && w < p.dwell_n ? p.dwell_rows + w * d : nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 tr, ar, wr, o;
            zero(tr); zero(ar); zero(wr);
            if (trow != nullptr) load_tile(tr, trow, nt, h);
            if (arow != nullptr) load_tile(ar, arow, nt, h);
            if (wrow != nullptr) load_tile(wr, wrow, nt, h);
            if (p.pre != nullptr) {
                f32x16 pr;
                load_tile(pr, p.pre + tt * d, nt, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] += pr[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = (((acc[nt][r] * p.scale + tr[r]) + ar[r]) + wr[r]) * pm;
            if (ok) store_tile(o, p.out + t * d, nt, h);
        }
        if (ok && h == 0) {
            p.mask[t] = pm;
            p.pos_idx[t] = pos;
            p.time_bin[t] = tb;
        }
    }
}

// g = dout * pm, gs = g * scale, de = Wh[:, :ds]^T gs  (MT = ds / 32 output tiles)
template <int NT>
__global__ __launch_bounds__(256) void input_assembly_bwd_kernel(AssemblyParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int d = 32 * NT, MT = p.ds / 32, KS = 2 * NT;
    weight_frags_to_lds(p.W, 1, p.ldw, MT, KS);          // transposed view: (ds x d)
    const WSrc<true> src{reinterpret_cast<const uint4*>(p.W), 0u, (unsigned)(MT * KS * 128)};
    const int64_t n_tiles = hmx_ceil_div(p.T, 32);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t t = tile * 32 + i;
        const bool ok = t < p.T;
        const int64_t tt = ok ? t : p.T - 1;
        const float pm = ok ? p.pmask[tt] : 0.f;
        Frag gf[2 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 g, gs;
            load_tile(g, p.dout + tt * d, nt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g[r] *= pm;
                gs[r] = g[r] * p.scale;
            }
            if (ok) {
                store_tile(g, p.g + t * d, nt, h);
                store_tile(gs, p.gs + t * d, nt, h);
            }
            tile_frags(gs, gf[2 * nt], gf[2 * nt + 1]);
        }
#pragma unroll 1
        for (int mt = 0; mt < MT; ++mt) {
            f32x16 acc;
            zero(acc);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks) mma<true>(acc, src, 0, mt, KS, ks, gf[ks], lane);
            if (ok) store_tile(acc, p.de + t * p.ds, mt, h);
        }
    }
}

// workgroups of a launch whose workgroups hold `waves` waves (one 32-token tile per wave and trip)
inline int block_grid(int64_t T, int waves = 4) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(hmx_ceil_div(hmx_ceil_div(T, 32), waves), 8192 / waves));
}

bool shape_ok(int64_t d, int64_t H) {
    if (H <= 0 || d <= 0 || d % H) return false;
    const int64_t hd = d / H;
    return d % 32 == 0 && d <= 32 * kMaxNT && hd % 32 == 0 && H <= 8;
}

constexpr size_t kMaxWeightLds = 128 * 1024;

// Launch KERNEL<NT, LDSW>.  With the entry range [lo, hi) non-empty and small enough, every workgroup keeps it in LDS
// (the copy is paid once per CU: one workgroup per CU walks the token tiles); otherwise weights come from L2.
// Measured at d = 64: the backward halves (512 registers, one wave per SIMD either way) gain 7-10 % from LDS weights;
// the forward halves fit two waves per SIMD without the LDS footprint and are faster that way, so they pass lo = hi.
#define HMX_BLOCK_LAUNCH(KERNEL, NTV, WAVES, BWD, p, T, lo, hi, stream)                                                   \
    do {                                                                                                             \
        const size_t lds_bytes = ((hi) - (lo)) * sizeof(uint4);                                                      \
        constexpr bool kCanLds = (NTV) < 4;      /* one d = 128 weight image is 64 KB: no useful range fits */        \
        if (kCanLds && (hi) > (lo) && lds_bytes <= kMaxWeightLds) {                                                  \
            (p).lds_lo = (unsigned)(lo);                                                                             \
            (p).lds_hi = (unsigned)(hi);                                                                             \
            hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<NTV, kCanLds>), (int)kMaxWeightLds);         \
            hipLaunchKernelGGL((KERNEL<NTV, kCanLds>), dim3(std::min(block_grid(T, WAVES), 256)), dim3(64 * (WAVES)), \
                               lds_bytes, stream, p);                                                                \
        } else {                                                                                                     \
            (p).lds_lo = (p).lds_hi = 0;                                                                             \
            const size_t sums_bytes = (BWD) && sums_in_lds(NTV, false) ? kSumsLdsBytes : 0;                          \
            if (sums_bytes) hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<NTV, false>), (int)sums_bytes); \
            hipLaunchKernelGGL((KERNEL<NTV, false>), dim3(std::min(block_grid(T, WAVES), 4096 / (WAVES))),           \
                               dim3(64 * (WAVES)), sums_bytes, stream, p);                                           \
        }                                                                                                            \
    } while (0)

// BWD: the kernel is a backward half (bwd_waves(NT) waves per workgroup); forward halves have four
#define HMX_BLOCK_DISPATCH(KERNEL, BWD, p, lo, hi, stream)                                                           \
    switch ((p).lay.NT) {                                                                                            \
        case 1: HMX_BLOCK_LAUNCH(KERNEL, 1, (BWD) ? bwd_waves(1) : 4, BWD, p, (p).T, lo, hi, stream); break;              \
        case 2: HMX_BLOCK_LAUNCH(KERNEL, 2, (BWD) ? bwd_waves(2) : 4, BWD, p, (p).T, lo, hi, stream); break;              \
        case 3: HMX_BLOCK_LAUNCH(KERNEL, 3, (BWD) ? bwd_waves(3) : 4, BWD, p, (p).T, lo, hi, stream); break;              \
        default: HMX_BLOCK_LAUNCH(KERNEL, 4, (BWD) ? bwd_waves(4) : 4, BWD, p, (p).T, lo, hi, stream); break;             \
    }

}  // namespace

// ================================================================================ C ABI
extern "C" int hmx_block_supported(int64_t d, int64_t H) { return shape_ok(d, H) ? 1 : 0; }

// Width the fused halves run a (d, H) block at: every head widened with zero features to the next multiple of 32
// (d itself when d / H already is one); 0 when that width is not supported or head_dim is odd.
extern "C" int64_t hmx_block_padded_width(int64_t d, int64_t H) {
    if (H <= 0 || d <= 0 || d % H || (d / H) % 2) return 0;
    const int64_t hdp = (d / H + 31) / 32 * 32, dp = hdp * H;
    const int64_t EDp = 4 * hdp;
    if (!shape_ok(dp, H) || (EDp & (EDp - 1))) return 0;     // the scan groups 4 * head_dim channels in powers of two
    return dp;
}

extern "C" size_t hmx_block_packed_bytes(int64_t d, int64_t H) {
    if (!shape_ok(d, H)) return 0;
    return make_layout((int)d, (int)H).total * sizeof(uint4);
}

// upper bound of the waves of a backward launch (each writes one row of partials): whole workgroups of up to eight
extern "C" int64_t hmx_block_scale_partial_rows(int64_t T) {
    return std::min<int64_t>(hmx_ceil_div(hmx_ceil_div(std::max<int64_t>(T, 1), 32), 8) * 8, 4096);
}

extern "C" int hmx_block_pack_f32(const hmx_block_weights* w, int64_t d, int64_t H, int64_t d_real, void* packed,
                                  void* stream) {
    HMX_ENTER();
    if (!w || !packed || !shape_ok(d, H)) return HMX_EINVAL;
    if (!w->W1 || !w->W2 || !w->Wg || !w->Wu || !w->Wd) return HMX_EINVAL;
    if (d_real <= 0) d_real = d;
    const bool widened = d_real != d;
    if (widened && (d_real % H || hmx_block_padded_width(d_real, H) != d)) return HMX_EINVAL;
    if (widened && (!w->scale1 || !w->b1 || !w->b2 || !w->scale2)) return HMX_EINVAL;
    const PackLayout l = make_layout((int)d, (int)H);
    const int hd = (int)(d_real / H), ED = 4 * hd, D = (int)d_real;      // true sizes; l.* are the image's
    PackParams p{};
    int n = 0;
    // (rs, cs): strides of the TRUE weight; (N, K): image size; (gN -> gNp), (gK -> gKp): widened groups per axis
    auto add = [&](const float* W, int64_t rs, int64_t cs, int N, int K, int gN, int gNp, int gK, int gKp, size_t out) {
        p.job[n++] = PackJob{W, rs, cs, N, K, gN, gNp, gK, gKp, out};
    };
    const int Dp = (int)d;
    const float* sq[5] = {w->W1, w->W2, w->Wg, w->Wu, w->Wd};
    const size_t sq_out[5] = {l.w1, l.w2, l.wg, l.wu, l.wd}, sq_out_t[5] = {l.w1t, l.w2t, l.wgt, l.wut, l.wdt};
    for (int i = 0; i < 5; ++i) {
        add(sq[i], D, 1, Dp, Dp, hd, l.hd, hd, l.hd, sq_out[i]);
        add(sq[i], 1, D, Dp, Dp, hd, l.hd, hd, l.hd, sq_out_t[i]);
    }
    for (int hh = 0; hh < (int)H; ++hh) {
        if (!w->Win[hh] || !w->Wout[hh]) return HMX_EINVAL;
        add(w->Win[hh], hd, 1, 2 * l.ED, l.hd, ED, l.ED, hd, l.hd, l.win + hh * l.win_stride);       // (2 ED, hd)
        add(w->Win[hh], 1, hd, l.hd, 2 * l.ED, hd, l.hd, ED, l.ED, l.wint + hh * l.wint_stride);     // transposed
        add(w->Wout[hh], ED, 1, l.hd, l.ED, hd, l.hd, ED, l.ED, l.wout + hh * l.wout_stride);        // (hd, ED)
        add(w->Wout[hh], 1, ED, l.ED, l.hd, ED, l.ED, hd, l.hd, l.woutt + hh * l.woutt_stride);      // transposed
    }
    if (widened) {
        const float* vec[4] = {w->scale1, w->b1, w->b2, w->scale2};
        for (int i = 0; i < 4; ++i) add(vec[i], 0, 0, Dp, 1, hd, l.hd, 0, 0, l.vec + (size_t)i * (Dp / 4));
    }
    p.n_jobs = n;
    p.out = static_cast<uint4*>(packed);
    hipLaunchKernelGGL(block_pack_kernel, dim3(8, n), dim3(256), 0, (hipStream_t)stream, p);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

static bool fill_common(BlockParams& p, const hmx_block_args* a) {
    if (!a || !shape_ok(a->d, a->H) || a->T < 0 || a->L <= 0 || !a->packed) return false;
    p = BlockParams{};
    p.T = a->T; p.L = (int)a->L; p.first = a->first; p.eps = a->eps;
    p.inv_sqrt_h = 1.0f / sqrtf((float)a->H);
    const int64_t d_real = a->d_real > 0 ? a->d_real : a->d;
    if (d_real != a->d && hmx_block_padded_width(d_real, a->H) != a->d) return false;
    p.inv_d = 1.0f / (float)d_real;
    p.lay = make_layout((int)a->d, (int)a->H);
    p.pk = static_cast<const uint4*>(a->packed);
    p.mask = a->mask; p.x = a->x; p.drop = a->drop; p.scale1 = a->scale1; p.b1 = a->b1; p.cs = a->rope_cos; p.sn = a->rope_sin;
    p.rstd1 = a->rstd1; p.x_in = a->x_in; p.z = a->z; p.y2 = a->y2; p.yg = a->yg; p.b2 = a->b2; p.scale2 = a->scale2;
    p.out = a->out; p.m = a->m; p.rstd2 = a->rstd2; p.dout = a->dout; p.dyg = a->dyg; p.dy2 = a->dy2; p.dm = a->dm;
    p.hmid = a->hmid; p.n2 = a->n2; p.dg = a->dg; p.du = a->du; p.p = a->p; p.dy1 = a->dy1; p.dscale_part = a->dscale_part;
    p.scale_rows = (int)hmx_block_scale_partial_rows(a->T);
    p.dx_in = a->dx_in; p.dz = a->dz; p.dx = a->dx; p.s = a->s; p.dx1 = a->dx1; p.n = a->n;
    if (d_real != a->d) {       // widened layout: the d-vectors come from the pack buffer (zero in the added features)
        const float* vec = reinterpret_cast<const float*>(p.pk + p.lay.vec);
        p.scale1 = vec; p.b1 = vec + a->d; p.b2 = vec + 2 * a->d; p.scale2 = vec + 3 * a->d;
    }
    return true;
}

extern "C" int hmx_block_pre_fwd(const hmx_block_args* a, void* stream) {
    HMX_ENTER();
    BlockParams p;
    if (!fill_common(p, a)) return HMX_EINVAL;
    if (p.T == 0) return HMX_OK;
    if (!p.x || !p.scale1 || !p.b1 || !p.cs || !p.sn || !p.rstd1 || !p.x_in || !p.z || !p.y2) return HMX_EINVAL;
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    HMX_BLOCK_DISPATCH(block_pre_fwd_kernel, false, p, (size_t)0, (size_t)0, (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_block_post_fwd(const hmx_block_args* a, void* stream) {
    HMX_ENTER();
    BlockParams p;
    if (!fill_common(p, a)) return HMX_EINVAL;
    if (p.T == 0) return HMX_OK;
    if (!p.yg || !p.y2 || !p.x || !p.b2 || !p.scale2 || !p.out || !p.m || !p.rstd2 || (p.first && (!p.rstd1 || !p.scale1)))
        return HMX_EINVAL;
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    HMX_BLOCK_DISPATCH(block_post_fwd_kernel, false, p, (size_t)0, (size_t)0, (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_block_post_bwd(const hmx_block_args* a, void* stream) {
    HMX_ENTER();
    BlockParams p;
    if (!fill_common(p, a)) return HMX_EINVAL;
    if (p.T == 0) return HMX_OK;
    if (!p.dout || !p.m || !p.rstd2 || !p.scale2 || !p.yg || !p.y2 || !p.dyg || !p.dy2 || !p.dm || !p.hmid || !p.n2 || !p.dg ||
        !p.du || !p.p || !p.dy1 || !p.dscale_part)
        return HMX_EINVAL;
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    HMX_BLOCK_DISPATCH(block_post_bwd_kernel, true, p, p.lay.wg, p.lay.w2t + pack_entries(p.lay.d, p.lay.d), (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_block_pre_bwd(const hmx_block_args* a, void* stream) {
    HMX_ENTER();
    BlockParams p;
    if (!fill_common(p, a)) return HMX_EINVAL;
    if (p.T == 0) return HMX_OK;
    if (!p.x || !p.scale1 || !p.b1 || !p.cs || !p.sn || !p.dx_in || !p.dz || !p.dy2 || !p.dm || !p.dx || !p.s || !p.dx1 || !p.n ||
        !p.dscale_part)
        return HMX_EINVAL;
    void* span = hmx_profile_begin("proj", (hipStream_t)stream);
    HMX_BLOCK_DISPATCH(block_pre_bwd_kernel, true, p, p.lay.w1, p.lay.wint + p.lay.wint_stride * p.lay.H, (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
static bool assembly_shape_ok(const hmx_input_assembly_args* a) {
    if (!a || a->T < 0 || a->L <= 0 || a->d <= 0 || a->ds <= 0) return false;
    if (a->d % 32 || a->d > 32 * kMaxNT || a->ds % 32 || a->ds > 256) return false;
    if ((size_t)(a->d / 32) * (a->ds / 16) * 128 * sizeof(uint4) > kMaxWeightLds) return false;
    return a->W != nullptr && a->ldw >= a->ds;
}

extern "C" int hmx_input_assembly_supported(int64_t d, int64_t ds) {
    hmx_input_assembly_args a{};
    a.T = 1; a.L = 1; a.d = d; a.ds = ds; a.ldw = ds;
    a.W = reinterpret_cast<const float*>(16);
    return assembly_shape_ok(&a) ? 1 : 0;
}

static void fill_assembly(AssemblyParams& p, const hmx_input_assembly_args* a) {
    p = AssemblyParams{};
    p.T = a->T; p.L = (int)a->L; p.ds = (int)a->ds; p.d = (int)a->d; p.time_span = (int)a->time_span;
    p.ids = a->ids; p.token_type = a->token_type; p.ts = a->timestamps; p.action = a->action; p.dwell = a->dwell;
    p.item_table = a->item_table; p.user_table = a->user_table; p.pos_table = a->pos_table;
    p.item_rows = a->item_rows; p.user_rows = a->user_rows; p.pos_rows = a->pos_rows;
    p.W = a->W; p.ldw = a->ldw; p.bias = a->bias; p.pre = a->pre; p.scale = a->scale;
    p.time_rows = a->time_rows; p.action_rows = a->action_rows; p.dwell_rows = a->dwell_rows;
    p.time_n = a->time_n; p.action_n = a->action_n; p.dwell_n = a->dwell_n;
    p.out = a->out; p.mask = a->mask; p.pos_idx = a->pos_idx; p.time_bin = a->time_bin; p.e = a->e;
    p.dout = a->dout; p.pmask = a->mask; p.g = a->g; p.gs = a->gs; p.de = a->de;
}

#define HMX_ASSEMBLY_LAUNCH(KERNEL, p, lds, stream)                                                                  \
    do {                                                                                                             \
        const dim3 grid(std::min(block_grid((p).T, 4), 1024));                                                       \
        switch ((p).d / 32) {                                                                                        \
            case 1:                                                                                                  \
                hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<1>), (int)kMaxWeightLds);                \
                hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), lds, stream, p);                                    \
                break;                                                                                               \
            case 2:                                                                                                  \
                hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<2>), (int)kMaxWeightLds);                \
                hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), lds, stream, p);                                    \
                break;                                                                                               \
            case 3:                                                                                                  \
                hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<3>), (int)kMaxWeightLds);                \
                hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), lds, stream, p);                                    \
                break;                                                                                               \
            default:                                                                                                 \
                hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&KERNEL<4>), (int)kMaxWeightLds);                \
                hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), lds, stream, p);                                    \
                break;                                                                                               \
        }                                                                                                            \
    } while (0)

extern "C" int hmx_input_assembly_fwd(const hmx_input_assembly_args* a, void* stream) {
    HMX_ENTER();
    if (!assembly_shape_ok(a)) return HMX_EINVAL;
    if (a->T == 0) return HMX_OK;
    if (!a->ids || !a->token_type || !a->timestamps || !a->action || !a->dwell || !a->item_table || !a->user_table ||
        !a->pos_table || !a->bias || !a->time_rows || !a->action_rows || !a->dwell_rows || !a->out || !a->mask || !a->pos_idx ||
        !a->time_bin)
        return HMX_EINVAL;
    if (a->item_rows <= 0 || a->user_rows <= 0 || a->pos_rows < a->L + 1 || a->time_n <= 0 || a->action_n <= 0 ||
        a->dwell_n <= 0 || a->time_span <= 0)
        return HMX_EINVAL;
    const uintptr_t align = reinterpret_cast<uintptr_t>(a->item_table) | reinterpret_cast<uintptr_t>(a->user_table) |
                            reinterpret_cast<uintptr_t>(a->pos_table) | reinterpret_cast<uintptr_t>(a->bias) |
                            reinterpret_cast<uintptr_t>(a->time_rows) | reinterpret_cast<uintptr_t>(a->action_rows) |
                            reinterpret_cast<uintptr_t>(a->dwell_rows) | reinterpret_cast<uintptr_t>(a->out) |
                            reinterpret_cast<uintptr_t>(a->pre) | reinterpret_cast<uintptr_t>(a->e);
    if (align % 16) return HMX_EINVAL;
    AssemblyParams p;
    fill_assembly(p, a);
    const size_t lds = (size_t)(a->d / 32) * (a->ds / 16) * 128 * sizeof(uint4);
    void* span = hmx_profile_begin("input_assembly", (hipStream_t)stream);
    HMX_ASSEMBLY_LAUNCH(input_assembly_fwd_kernel, p, lds, (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_input_assembly_bwd(const hmx_input_assembly_args* a, void* stream) {
    HMX_ENTER();
    if (!assembly_shape_ok(a)) return HMX_EINVAL;
    if (a->T == 0) return HMX_OK;
    if (!a->dout || !a->mask || !a->g || !a->gs || !a->de) return HMX_EINVAL;
    const uintptr_t align = reinterpret_cast<uintptr_t>(a->dout) | reinterpret_cast<uintptr_t>(a->g) |
                            reinterpret_cast<uintptr_t>(a->gs) | reinterpret_cast<uintptr_t>(a->de);
    if (align % 16) return HMX_EINVAL;
    AssemblyParams p;
    fill_assembly(p, a);
    const size_t lds = (size_t)(a->ds / 32) * (a->d / 16) * 128 * sizeof(uint4);
    void* span = hmx_profile_begin("input_assembly", (hipStream_t)stream);
    HMX_ASSEMBLY_LAUNCH(input_assembly_bwd_kernel, p, lds, (hipStream_t)stream);
    hmx_profile_end(span, (hipStream_t)stream);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
