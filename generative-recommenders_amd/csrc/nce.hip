// Fused InfoNCE loss over [positive | hard negatives gathered from the item table | in-batch negatives]
// (hydra3.py:845-974): the (M, 1 + n_hard + n_inb, d) key tensor of the reference is never materialised.
//
// Per query row i:   logit[c] = <q_i / |q_i|, key_c / |key_c|> / temperature,   loss_i = logsumexp(logit) - logit[0]
//   key_0          = k[i]                       (its positive)
//   key_1..n_hard  = table[hard_ids[i, .]]      (row gather from the item table, normalised on the fly)
//   key_rest       = k[inb_idx[i, .]]           (other rows of the same batch)
//
// Geometry: one wave per query; four lanes x 16 B x ceil(d / 16) chunks cover one key row (d <= 256), so a wave
// handles sixteen keys per step.  Row sums use DPP inside a quad (no LDS).
// Backward: the same traversal yields d q and one coefficient per (query, key); the key gradients are the
// transposed sparse product  sum_i coef[i, c] * q_i  accumulated per destination row with the deterministic
// sorted-run sum shared with the embedding backward, followed by the normalisation Jacobian applied once per
// destination row (it is linear, so it commutes with the sum).
#include "segment_sum.h"
#include "profile.h"

namespace {

using namespace hmx_seg;

constexpr float kNormEps = 1e-12f;   // F.normalize(eps=1e-12)

struct NceParams {
    const float* q;           // (m, d)
    const float* k;           // (m, d)
    int64_t m;
    int d, cpr;               // cpr = d / 4
    const float* table;       // (n_rows, d)
    int64_t n_rows;
    const int64_t* hard_ids;  // (m, hard_ld): the first min(n_hard, hard_ld) hard slots of every row
    int n_hard, hard_ld;
    const int64_t* inb_idx;   // (m, n_inb) int64, or null when inb32 is given
    const int32_t* inb32;     // (m, n_inb) int32
    int inb_skip_self;        // inb32 holds draws r in [0, m - 1): the key is row r + (r >= i)
    int n_inb;
    float inv_temp;
    int normalize;
    float* logits;            // (m, C): logits after forward, key-gradient coefficients after backward
    float* row_loss;          // (m)
    float* lse;               // (m)
    float* q_inv;             // (m) 1 / max(|q_i|, eps)   (1 when normalize == 0)
    float* k_inv;             // (m)
    const float* grad_row_loss;   // (m)   backward only
    float* dq;                    // (m, d) backward only
    float* dq_unit;               // (m, d) or null.  Forward: receives d row_loss[i] / d q[i] (the gradient for a unit
                                  // upstream gradient), accumulated during the one key walk; backward: its source
    int period;                   // > 0: hard_ids[i, s] == hard_ids[i, s % period]; only the first `period` hard slots
                                  // are evaluated, each weighted by the number of slots it stands for
};

// The slots a kernel walks: all C of them, or with periodic hard negatives the positive, the first `period` hard
// slots and the in-batch slots.  real() maps a walked slot to its column of the logits matrix, copies() is how many
// columns it stands for.
struct SlotWalk {
    int n_hard, kh, n_walk;
    __device__ __forceinline__ explicit SlotWalk(const NceParams& p)
        : n_hard(p.n_hard), kh(p.period > 0 ? p.period : p.n_hard), n_walk(1 + kh + p.n_inb) {}
    __device__ __forceinline__ int real(int v) const { return v <= kh ? v : v - kh + n_hard; }
    __device__ __forceinline__ float copies(int v) const {
        return (v >= 1 && v <= kh && kh != n_hard) ? (float)((n_hard - v + kh) / kh) : 1.0f;
    }
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row receives the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0x128>(v);   // row_ror:8
    v += dpp_f32<0x124>(v);   // row_ror:4
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// one group of 16 lanes per row: out[i] = 1 / max(|x_i|, eps)
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ x, int64_t m, int cpr, int normalize,
                                                       float* __restrict__ out) {
    const int lc = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    float ss = 0.f;
    if (i < m && normalize)
        for (int c = lc; c < cpr; c += 16) {
            const float4 v = reinterpret_cast<const float4*>(x)[i * cpr + c];
            ss += dot4(v, v);
        }
    ss = row16_sum(ss);
    if (i < m && lc == 0) out[i] = normalize ? 1.0f / fmaxf(sqrtf(ss), kNormEps) : 1.0f;
}

// Where key `slot` of query i lives.  kind: 0 = absent (zero key), 1 = row of k (inverse norm known), 2 = table row.
struct KeyRef {
    const float4* src;
    float inv;
    int kind;
};
__device__ __forceinline__ KeyRef locate_key(const NceParams& p, int64_t i, int slot, int C) {
    KeyRef r{nullptr, 0.f, 0};
    if (slot >= C) return r;
    if (slot == 0) {
        r.src = reinterpret_cast<const float4*>(p.k) + i * p.cpr;
        r.inv = p.k_inv[i];
        r.kind = 1;
    } else if (slot <= p.n_hard) {
        const int64_t id = p.hard_ids[i * p.hard_ld + slot - 1];
        if (id >= 0 && id < p.n_rows) {
            r.src = reinterpret_cast<const float4*>(p.table) + id * p.cpr;
            r.kind = 2;
        }
    } else {
        const int64_t at = i * p.n_inb + slot - 1 - p.n_hard;
        int64_t j;
        if (p.inb32 != nullptr) {
            j = p.inb32[at];
            if (p.inb_skip_self) j += (j >= i) ? 1 : 0;
        } else {
            j = p.inb_idx[at];
        }
        if (j >= 0 && j < p.m) {
            r.src = reinterpret_cast<const float4*>(p.k) + j * p.cpr;
            r.inv = p.k_inv[j];
            r.kind = 1;
        }
    }
    return r;
}

// Geometry of the two key walks: one wave per query; a key row is read by FOUR lanes (a DPP quad), each taking every
// fourth 16-byte chunk, so a wave handles sixteen keys per step and the per-key bookkeeping (slot -> address, the
// soft-max update, the coefficient) is issued once per sixteen keys.  NCH = chunks per lane = ceil(d / 16).
// sum over the 4 lanes of a quad; every lane of the quad receives the total
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    return v;
}
template <int NCH>
struct WalkSteps {
    static constexpr int value = NCH <= 2 ? 4 : (NCH == 4 ? 2 : 1);   // steps of 16 keys whose loads are in flight together
};

template <int NCH, bool TRAIN>
__global__ __launch_bounds__(256) void nce_fwd_kernel(NceParams p) {
    constexpr int kSteps = WalkSteps<NCH>::value;
    const int lane = threadIdx.x & 63, kg = lane >> 2, sub = lane & 3;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.m) return;   // wave-uniform
    const int C = 1 + p.n_hard + p.n_inb;
    const SlotWalk walk(p);
    float4 qv[NCH];
    const float qs = p.q_inv[i] * p.inv_temp;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = sub + 4 * ch;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.cpr) v = reinterpret_cast<const float4*>(p.q)[i * p.cpr + c];
        qv[ch] = make_float4(v.x * qs, v.y * qs, v.z * qs, v.w * qs);
    }
    float mx = -INFINITY, sm = 0.f, l0 = 0.f;
    // TRAIN: acc = sum over this quad's keys of copies * exp(logit - mx) * key / |key| (the soft-max weighted key sum
    // of the query's gradient), kept on the same running maximum as sm
    float4 acc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) acc[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < walk.n_walk; c0 += 16 * kSteps) {
        KeyRef ref[kSteps];
        float4 v[kSteps][NCH];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int vs = c0 + 16 * u + kg;
            ref[u] = locate_key(p, i, vs < walk.n_walk ? walk.real(vs) : C, C);
        }
#pragma unroll
        for (int u = 0; u < kSteps; ++u)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = sub + 4 * ch;
                v[u][ch] = (ref[u].kind != 0 && c < p.cpr) ? ref[u].src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int vs = c0 + 16 * u + kg;
            const int slot = vs < walk.n_walk ? walk.real(vs) : C;
            float dot = 0.f, ss = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                dot += dot4(v[u][ch], qv[ch]);
                ss += dot4(v[u][ch], v[u][ch]);
            }
            dot = quad_sum(dot);
            float inv = ref[u].inv;
            if (ref[u].kind == 2) {
                ss = quad_sum(ss);
                inv = p.normalize ? 1.0f / fmaxf(sqrtf(ss), kNormEps) : 1.0f;
            }
            const float logit = dot * inv;
            if (slot < C) {
                if (sub == 0) p.logits[i * C + slot] = logit;
                // running maximum, rescaled only when it moves (a handful of times per query): one hardware
                // exponential per key instead of two library ones.  exp(-inf) = 0 on the first key.
                if (logit > mx) {
                    const float down = __expf(mx - logit);
                    sm *= down;
                    if (TRAIN) {
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) acc[ch] = vscale(down, acc[ch]);
                    }
                    mx = logit;
                }
                const float w = walk.copies(vs) * __expf(logit - mx);
                sm += w;
                if (TRAIN) {
                    const float wk = w * inv;
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        acc[ch].x += wk * v[u][ch].x;
                        acc[ch].y += wk * v[u][ch].y;
                        acc[ch].z += wk * v[u][ch].z;
                        acc[ch].w += wk * v[u][ch].w;
                    }
                }
                if (slot == 0) l0 = logit;
            }
        }
    }
    // the sixteen key groups of the wave (every lane of a quad holds the same pair)
#pragma unroll
    for (int off = 4; off <= 32; off <<= 1) {
        const float om = __shfl_xor(mx, off), os = __shfl_xor(sm, off);
        const float nm = fmaxf(mx, om);
        const float fa = (mx == -INFINITY) ? 0.f : __expf(mx - nm);
        const float fb = (om == -INFINITY) ? 0.f : __expf(om - nm);
        sm = sm * fa + os * fb;
        if (TRAIN) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[ch].x = acc[ch].x * fa + __shfl_xor(acc[ch].x, off) * fb;
                acc[ch].y = acc[ch].y * fa + __shfl_xor(acc[ch].y, off) * fb;
                acc[ch].z = acc[ch].z * fa + __shfl_xor(acc[ch].z, off) * fb;
                acc[ch].w = acc[ch].w * fa + __shfl_xor(acc[ch].w, off) * fb;
            }
        }
        mx = nm;
    }
    l0 = __shfl(l0, 0);
    if (lane == 0) {
        const float lse = mx + logf(sm);
        p.lse[i] = lse;
        p.row_loss[i] = lse - l0;
    }
    if (TRAIN) {
        // d row_loss / d qn = (sum_c softmax_c keyn_c - keyn_0) / temperature; through qn = q / max(|q|, eps) as in
        // nce_bwd_q_kernel.  Every quad holds the wave's totals; the first one writes.
        const float qinv = p.q_inv[i], kinv = p.k_inv[i], rs = 1.0f / sm, temp = 1.0f / p.inv_temp;
        float4 qn[NCH];
        float pr = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = sub + 4 * ch;
            float4 k0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < p.cpr) k0 = reinterpret_cast<const float4*>(p.k)[i * p.cpr + c];
            acc[ch].x = (acc[ch].x * rs - k0.x * kinv) * p.inv_temp;
            acc[ch].y = (acc[ch].y * rs - k0.y * kinv) * p.inv_temp;
            acc[ch].z = (acc[ch].z * rs - k0.z * kinv) * p.inv_temp;
            acc[ch].w = (acc[ch].w * rs - k0.w * kinv) * p.inv_temp;
            qn[ch] = vscale(temp, qv[ch]);
            pr += dot4(qn[ch], acc[ch]);
        }
        pr = quad_sum(pr);
        if (!p.normalize || qinv >= 1.0f / kNormEps) pr = 0.f;
        if (kg == 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = sub + 4 * ch;
                if (c < p.cpr) {
                    float4 o;
                    o.x = qinv * (acc[ch].x - qn[ch].x * pr);
                    o.y = qinv * (acc[ch].y - qn[ch].y * pr);
                    o.z = qinv * (acc[ch].z - qn[ch].z * pr);
                    o.w = qinv * (acc[ch].w - qn[ch].w * pr);
                    reinterpret_cast<float4*>(p.dq_unit)[i * p.cpr + c] = o;
                }
            }
        }
    }
}

// Backward without a key walk, for a forward pass that left dq_unit: dq[i] = grad_row_loss[i] * dq_unit[i], and the
// walked columns of `logits` become the key-gradient coefficients (what nce_bwd_q_kernel leaves there):
//   coef[i, c] = (softmax_c - [c == 0]) * copies(c) * grad_row_loss[i] / temperature * q_inv[i].
// One wave per query row.
__global__ __launch_bounds__(256) void nce_bwd_light_kernel(NceParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.m) return;
    const int C = 1 + p.n_hard + p.n_inb;
    const SlotWalk walk(p);
    const float gr = p.grad_row_loss[i], lse = p.lse[i];
    const float g = gr * p.inv_temp * p.q_inv[i];
    float* row = p.logits + i * C;
    // the row's logits are requested eight chunks (512 slots) at a time before the first is used: a wave lives for
    // one round trip of its row instead of one per chunk
    const float4 u = lane < p.cpr ? reinterpret_cast<const float4*>(p.dq_unit)[i * p.cpr + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v0 = 0; v0 < walk.n_walk; v0 += 8 * 64) {
        float lg[8];
        int slot[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int vs = v0 + 64 * k + lane;
            slot[k] = vs < walk.n_walk ? walk.real(vs) : -1;
            lg[k] = slot[k] >= 0 ? row[slot[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int vs = v0 + 64 * k + lane;
            if (slot[k] >= 0) row[slot[k]] = (__expf(lg[k] - lse) - (slot[k] == 0 ? 1.f : 0.f)) * g * walk.copies(vs);
        }
    }
    if (lane < p.cpr) reinterpret_cast<float4*>(p.dq)[i * p.cpr + lane] = vscale(gr, u);
    for (int c = lane + 64; c < p.cpr; c += 64) {
        const float4 w = reinterpret_cast<const float4*>(p.dq_unit)[i * p.cpr + c];
        reinterpret_cast<float4*>(p.dq)[i * p.cpr + c] = vscale(gr, w);
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void nce_bwd_q_kernel(NceParams p) {
    constexpr int kSteps = WalkSteps<NCH>::value;
    const int lane = threadIdx.x & 63, kg = lane >> 2, sub = lane & 3;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.m) return;
    const int C = 1 + p.n_hard + p.n_inb;
    const SlotWalk walk(p);
    const float qinv = p.q_inv[i], lse = p.lse[i], g = p.grad_row_loss[i] * p.inv_temp;
    float4 qn[NCH], acc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = sub + 4 * ch;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.cpr) v = reinterpret_cast<const float4*>(p.q)[i * p.cpr + c];
        qn[ch] = make_float4(v.x * qinv, v.y * qinv, v.z * qinv, v.w * qinv);
        acc[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int c0 = 0; c0 < walk.n_walk; c0 += 16 * kSteps) {
        KeyRef ref[kSteps];
        float4 v[kSteps][NCH];
        float lg[kSteps];
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int vs = c0 + 16 * u + kg;
            const int slot = vs < walk.n_walk ? walk.real(vs) : C;
            ref[u] = locate_key(p, i, slot, C);
            lg[u] = slot < C ? p.logits[i * C + slot] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kSteps; ++u)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = sub + 4 * ch;
                v[u][ch] = (ref[u].kind != 0 && c < p.cpr) ? ref[u].src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            const int vs = c0 + 16 * u + kg;
            const int slot = vs < walk.n_walk ? walk.real(vs) : C;
            float inv = ref[u].inv;
            if (ref[u].kind == 2) {
                float ss = 0.f;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) ss += dot4(v[u][ch], v[u][ch]);
                ss = quad_sum(ss);
                inv = p.normalize ? 1.0f / fmaxf(sqrtf(ss), kNormEps) : 1.0f;
            }
            if (slot < C) {
                // dL / d <qn, keyn>, for all the slots this one stands for
                const float w = (__expf(lg[u] - lse) - (slot == 0 ? 1.f : 0.f)) * g * walk.copies(vs);
                const float wk = w * inv;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    acc[ch].x += wk * v[u][ch].x;
                    acc[ch].y += wk * v[u][ch].y;
                    acc[ch].z += wk * v[u][ch].z;
                    acc[ch].w += wk * v[u][ch].w;
                }
                // d keyn = w * qn = (w * qinv) * q_i: the coefficient of q_i in the destination row's sum
                if (sub == 0) p.logits[i * C + slot] = w * qinv;
            }
        }
    }
    // add the sixteen key groups' partial rows (lanes with the same position in their quad hold the same chunks)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int off = 4; off <= 32; off <<= 1) {
            acc[ch].x += __shfl_xor(acc[ch].x, off);
            acc[ch].y += __shfl_xor(acc[ch].y, off);
            acc[ch].z += __shfl_xor(acc[ch].z, off);
            acc[ch].w += __shfl_xor(acc[ch].w, off);
        }
    }
    // acc = dL / d qn.  Through qn = q / max(|q|, eps):  dq = qinv * (acc - qn <qn, acc>) when |q| > eps, else qinv * acc
    float pr = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) pr += dot4(qn[ch], acc[ch]);
    pr = quad_sum(pr);
    if (!p.normalize || qinv >= 1.0f / kNormEps) pr = 0.f;
    if (kg == 0) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = sub + 4 * ch;
            if (c < p.cpr) {
                float4 o;
                o.x = qinv * (acc[ch].x - qn[ch].x * pr);
                o.y = qinv * (acc[ch].y - qn[ch].y * pr);
                o.z = qinv * (acc[ch].z - qn[ch].z * pr);
                o.w = qinv * (acc[ch].w - qn[ch].w * pr);
                reinterpret_cast<float4*>(p.dq)[i * p.cpr + c] = o;
            }
        }
    }
}

// In place on one gradient row G that belongs to raw row x with inverse norm `inv`:
//   G <- inv * (G - xn <xn, G>),  xn = x * inv      (inv * G when the norm was clamped)
__device__ __forceinline__ void project_row(float4* g_row, const float4* x_row, float inv, float extra_coef,
                                            const float4* extra_row, int cpr, int lc, int normalize) {
    float4 g[4], xn[4];
    float pr = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int c = lc + 16 * ch;
        g[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
        xn[ch] = g[ch];
        if (c < cpr) {
            g[ch] = g_row[c];
            if (extra_row != nullptr) {
                const float4 e = extra_row[c];
                g[ch].x += extra_coef * e.x;
                g[ch].y += extra_coef * e.y;
                g[ch].z += extra_coef * e.z;
                g[ch].w += extra_coef * e.w;
            }
            const float4 x = x_row[c];
            xn[ch] = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
            pr += dot4(xn[ch], g[ch]);
        }
    }
    pr = row16_sum(pr);
    float s = inv;
    if (!normalize) {
        pr = 0.f;
        s = 1.f;
    } else if (inv >= 1.0f / kNormEps) {
        pr = 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int c = lc + 16 * ch;
        if (c < cpr) {
            float4 o;
            o.x = s * (g[ch].x - xn[ch].x * pr);
            o.y = s * (g[ch].y - xn[ch].y * pr);
            o.z = s * (g[ch].z - xn[ch].z * pr);
            o.w = s * (g[ch].w - xn[ch].w * pr);
            g_row[c] = o;
        }
    }
}

// dk[j] <- Jacobian of normalisation applied to (dk[j] + coef[j, 0] * q[j])
__global__ __launch_bounds__(256) void nce_finish_k_kernel(NceParams p, float* __restrict__ dk) {
    const int lc = threadIdx.x & 15;
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int C = 1 + p.n_hard + p.n_inb;
    const bool ok = j < p.m;
    const int64_t jj = ok ? j : 0;
    // every lane of a DPP row takes part in row16_sum, so out-of-range groups run on row 0 and skip the store
    if (ok)
        project_row(reinterpret_cast<float4*>(dk) + jj * p.cpr, reinterpret_cast<const float4*>(p.k) + jj * p.cpr,
                    p.k_inv[jj], p.logits[jj * C], reinterpret_cast<const float4*>(p.q) + jj * p.cpr, p.cpr, lc,
                    p.normalize);
}

// Table rows that received hard-negative gradient: one 16-lane group per position of the sorted key array acts
// where a run of equal keys starts, i.e. exactly once per distinct row.
__global__ __launch_bounds__(256) void nce_project_table_rows_kernel(const uint32_t* __restrict__ skeys, int64_t n,
                                                                     uint32_t sentinel, const float* __restrict__ table,
                                                                     int cpr, float* __restrict__ dtable) {
    const int lc = threadIdx.x & 15;
    for (int64_t e = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); e < n; e += (int64_t)gridDim.x * 16) {
        const uint32_t key = skeys[e];
        if (key == sentinel) break;   // sentinels sort last (uniform inside a 16-lane row; rows of a wave may differ)
        if (e > 0 && skeys[e - 1] == key) continue;
        const float4* x = reinterpret_cast<const float4*>(table) + (int64_t)key * cpr;
        float ss = 0.f;
        for (int c = lc; c < cpr; c += 16) {
            const float4 v = x[c];
            ss += dot4(v, v);
        }
        ss = row16_sum(ss);
        const float inv = 1.0f / fmaxf(sqrtf(ss), kNormEps);
        project_row(reinterpret_cast<float4*>(dtable) + (int64_t)key * cpr, x, inv, 0.f, nullptr, cpr, lc, 1);
    }
}

// Hard negatives with a repeating structure (hydra3.py:905-919 tiles each user's K mined ids over the n_hard slots of
// every position of that user): inside a group of consecutive query rows [group_start[g], group_start[g + 1]) slot s
// refers to the same table row as slot s % K of the group's first row.  The table gradient of the group is then a
// small dense product: out[g * K + u] = sum_i (sum_{s = u, u + K, ...} coef[i, 1 + s]) * q_i, summed over the
// group's rows in order -- K rows per group to scatter instead of n_hard per query.
constexpr int kMaxPeriod = 64;
__global__ __launch_bounds__(256) void nce_group_rows_kernel(const float* __restrict__ q, const float* __restrict__ coef,
                                                             int64_t m, int C, int n_hard, int hard_ld, int K, int summed,
                                                             const int64_t* __restrict__ group_start,
                                                             const int64_t* __restrict__ hard_ids, int cpr,
                                                             float* __restrict__ out_rows, int64_t* __restrict__ out_ids) {
    constexpr int UMAX = kMaxPeriod / 16;
    const int lc = threadIdx.x & 15, lg = threadIdx.x >> 4;
    const int64_t g = blockIdx.x;
    int64_t r0 = group_start[g], r1 = group_start[g + 1];
    r0 = r0 < 0 ? 0 : (r0 > m ? m : r0);
    r1 = r1 < r0 ? r0 : (r1 > m ? m : r1);
    if (threadIdx.x < K) out_ids[g * K + threadIdx.x] = (r1 > r0) ? hard_ids[r0 * hard_ld + threadIdx.x] : -1;
    for (int c = lc; c < cpr; c += 16) {
        float4 acc[UMAX];
#pragma unroll
        for (int uu = 0; uu < UMAX; ++uu) acc[uu] = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight rows of q and their coefficients are requested before the first is used (a row at a time the walk
        // was a chain of dependent round trips: 1.2 ms for 512 users of 512 positions); rows are still added in order
        constexpr int RB = 8;
        for (int64_t i0 = r0; i0 < r1; i0 += RB) {
            float4 qv[RB];
            float cs[RB][UMAX];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const bool in = i0 + j < r1;
                const int64_t i = in ? i0 + j : r1 - 1;
                qv[j] = reinterpret_cast<const float4*>(q)[i * cpr + c];
                const float* ci = coef + i * C + 1;
#pragma unroll
                for (int uu = 0; uu < UMAX; ++uu) {
                    const int u = lg + 16 * uu;
                    float v = 0.f;
                    if (u < K && in) {
                        if (summed) v = ci[u];            // the coefficient already stands for all repeats of the slot
                        else for (int sl = u; sl < n_hard; sl += K) v += ci[sl];
                    }
                    cs[j][uu] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int uu = 0; uu < UMAX; ++uu) {
                    acc[uu].x += cs[j][uu] * qv[j].x;
                    acc[uu].y += cs[j][uu] * qv[j].y;
                    acc[uu].z += cs[j][uu] * qv[j].z;
                    acc[uu].w += cs[j][uu] * qv[j].w;
                }
        }
#pragma unroll
        for (int uu = 0; uu < UMAX; ++uu) {
            const int u = lg + 16 * uu;
            if (u < K) reinterpret_cast<float4*>(out_rows)[(g * K + u) * cpr + c] = acc[uu];
        }
    }
}

bool nce_shape_ok(int64_t m, int64_t d, int64_t n_rows, int64_t n_hard, int64_t n_inb) {
    if (m <= 0 || d <= 0 || d % 4 != 0 || d > 256 || n_hard < 0 || n_inb < 0) return false;
    if (n_hard > 0 && n_rows <= 0) return false;
    if (m >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31)) return false;
    if (m * (n_hard > n_inb ? n_hard : n_inb) >= ((int64_t)1 << 32)) return false;
    if (1 + n_hard + n_inb > (1 << 20)) return false;
    return true;
}

NceParams make_params(const float* q, const float* k, int64_t m, int64_t d, const float* table, int64_t n_rows,
                      const int64_t* hard_ids, int64_t n_hard, const void* inb_idx, int64_t n_inb, int64_t hard_ld,
                      int64_t inb_flags, float temperature, int normalize, float* logits, float* row_loss, float* lse,
                      float* q_inv, float* k_inv) {
    NceParams p{};
    p.q = q;
    p.k = k;
    p.m = m;
    p.d = (int)d;
    p.cpr = (int)(d / 4);
    p.table = table;
    p.n_rows = n_rows;
    p.hard_ids = hard_ids;
    p.n_hard = (int)n_hard;
    p.hard_ld = (int)(hard_ld > 0 ? hard_ld : n_hard);
    if (inb_flags & HMX_NCE_INB_I32) {
        p.inb32 = static_cast<const int32_t*>(inb_idx);
        p.inb_skip_self = (inb_flags & HMX_NCE_INB_SKIP_SELF) ? 1 : 0;
    } else {
        p.inb_idx = static_cast<const int64_t*>(inb_idx);
    }
    p.n_inb = (int)n_inb;
    p.inv_temp = 1.0f / temperature;
    p.normalize = normalize;
    p.logits = logits;
    p.row_loss = row_loss;
    p.lse = lse;
    p.q_inv = q_inv;
    p.k_inv = k_inv;
    return p;
}

bool aligned16(const void* a) { return (reinterpret_cast<uintptr_t>(a) & 15) == 0; }

// hard_ld: row stride of hard_ids (0 = n_hard); it may be shorter than n_hard only when the slots beyond it repeat
// (period <= hard_ld).  skip-self draws need int32 indices.
bool index_args_ok(int64_t n_hard, int64_t hard_ld, int64_t period, int64_t inb_flags) {
    if (hard_ld < 0 || (inb_flags & ~(int64_t)(HMX_NCE_INB_I32 | HMX_NCE_INB_SKIP_SELF))) return false;
    if ((inb_flags & HMX_NCE_INB_SKIP_SELF) && !(inb_flags & HMX_NCE_INB_I32)) return false;
    if (hard_ld > 0 && hard_ld < (period > 0 ? period : n_hard)) return false;
    return true;
}

}  // namespace

static int nce_fwd_impl(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                        int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, const void* inb_idx,
                        int64_t n_inb, int64_t hard_ld, int64_t inb_flags, int64_t period, float temperature, int normalize,
                        float* logits, float* row_loss, float* lse, float* q_inv, float* k_inv, float* dq_unit,
                        void* stream) {
    if (m == 0) return HMX_OK;
    if (period < 0 || period > n_hard || !index_args_ok(n_hard, hard_ld, period, inb_flags)) return HMX_EINVAL;
    if (!nce_shape_ok(m, d, n_rows, n_hard, n_inb) || !(temperature > 0.f)) return HMX_EINVAL;
    if (!q || !k || !logits || !row_loss || !lse || !q_inv || !k_inv) return HMX_EINVAL;
    if ((n_hard > 0 && (!table || !hard_ids)) || (n_inb > 0 && !inb_idx)) return HMX_EINVAL;
    if (!aligned16(q) || !aligned16(k) || (n_hard > 0 && !aligned16(table)) || !aligned16(dq_unit)) return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    NceParams p = make_params(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, temperature,
                              normalize, logits, row_loss, lse, q_inv, k_inv);
    p.period = (int)period;
    p.dq_unit = dq_unit;
    const unsigned g16 = (unsigned)hmx_ceil_div(m, 16), g4 = (unsigned)hmx_ceil_div(m, 4);
    hipLaunchKernelGGL(inv_norm_kernel, dim3(g16), dim3(256), 0, s, q, m, p.cpr, normalize, q_inv);
    hipLaunchKernelGGL(inv_norm_kernel, dim3(g16), dim3(256), 0, s, k, m, p.cpr, normalize, k_inv);
    void* span = hmx_profile_begin("nce_fwd", s);
#define HMX_NCE_FWD(NCH)                                                                                  \
    do {                                                                                                  \
        if (dq_unit) hipLaunchKernelGGL((nce_fwd_kernel<NCH, true>), dim3(g4), dim3(256), 0, s, p);       \
        else hipLaunchKernelGGL((nce_fwd_kernel<NCH, false>), dim3(g4), dim3(256), 0, s, p);              \
    } while (0)
    if (p.cpr <= 4) HMX_NCE_FWD(1);
    else if (p.cpr <= 8) HMX_NCE_FWD(2);
    else if (p.cpr <= 16) HMX_NCE_FWD(4);
    else if (p.cpr <= 32) HMX_NCE_FWD(8);
    else HMX_NCE_FWD(16);
#undef HMX_NCE_FWD
    hmx_profile_end(span, s);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_nce_fwd_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                               int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, const void* inb_idx,
                               int64_t n_inb, int64_t hard_ld, int64_t inb_flags, float temperature, int normalize,
                               float* logits, float* row_loss, float* lse, float* q_inv, float* k_inv, float* dq_unit,
                               void* stream) {
    HMX_ENTER();
    return nce_fwd_impl(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, 0, temperature,
                        normalize, logits, row_loss, lse, q_inv, k_inv, dq_unit, stream);
}

extern "C" int hmx_nce_fwd_periodic_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                                        int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, int64_t period,
                                        const void* inb_idx, int64_t n_inb, int64_t hard_ld, int64_t inb_flags,
                                        float temperature, int normalize, float* logits, float* row_loss, float* lse,
                                        float* q_inv, float* k_inv, float* dq_unit, void* stream) {
    HMX_ENTER();
    if (period <= 0) return HMX_EINVAL;
    return nce_fwd_impl(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, period, temperature,
                        normalize, logits, row_loss, lse, q_inv, k_inv, dq_unit, stream);
}

namespace {
size_t grouped_table_bytes(int64_t n_groups, int64_t period, int64_t n_rows, int64_t d) {
    const int64_t n = n_groups * period;
    return hmx_align_up((size_t)n * d * sizeof(float), 256) + hmx_align_up((size_t)n * sizeof(int64_t), 256) +
           hmx_seg::workspace_bytes<PlainRows>(n, n_rows, d);
}
}  // namespace

extern "C" size_t hmx_nce_bwd_workspace_bytes(int64_t m, int64_t d, int64_t n_rows, int64_t n_hard, int64_t n_inb) {
    if (!nce_shape_ok(m, d, n_rows, n_hard, n_inb)) return 256;
    size_t a = n_inb > 0 ? hmx_seg::workspace_bytes<WeightedRows>(m * n_inb, m, d) : 256;
    size_t b = n_hard > 0 ? hmx_seg::workspace_bytes<WeightedRows>(m * n_hard, n_rows, d) : 256;
    return a > b ? a : b;
}

extern "C" size_t hmx_nce_bwd_grouped_workspace_bytes(int64_t m, int64_t d, int64_t n_rows, int64_t n_hard, int64_t n_inb,
                                                      int64_t n_groups, int64_t period) {
    if (!nce_shape_ok(m, d, n_rows, n_hard, n_inb) || n_groups <= 0 || period <= 0) return 256;
    size_t a = n_inb > 0 ? hmx_seg::workspace_bytes<WeightedRows>(m * n_inb, m, d) : 256;
    size_t b = n_hard > 0 ? grouped_table_bytes(n_groups, period, n_rows, d) : 256;
    return a > b ? a : b;
}

static int nce_bwd_impl(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                        int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, const void* inb_idx,
                        int64_t n_inb, int64_t hard_ld, int64_t inb_flags, float temperature, int normalize, float* logits,
                        const float* lse,
                        const float* q_inv, const float* k_inv, const float* grad_row_loss, const float* dq_unit,
                        float* dq, float* dk, float* dtable, uint8_t* dtable_flags, const int64_t* group_start,
                        int64_t n_groups, int64_t period, void* workspace, size_t workspace_bytes, void* stream) {
    if (m == 0) return HMX_OK;
    const bool grouped = group_start != nullptr && n_hard > 0;
    if (grouped && (n_groups <= 0 || period <= 0 || period > kMaxPeriod || period > n_hard)) return HMX_EINVAL;
    if (!index_args_ok(n_hard, hard_ld, grouped ? period : 0, inb_flags)) return HMX_EINVAL;
    if (!nce_shape_ok(m, d, n_rows, n_hard, n_inb) || !(temperature > 0.f)) return HMX_EINVAL;
    if (!q || !k || !logits || !lse || !q_inv || !k_inv || !grad_row_loss || !dq || !dk || !workspace)
        return HMX_EINVAL;
    if ((n_hard > 0 && (!table || !hard_ids || !dtable)) || (n_inb > 0 && !inb_idx)) return HMX_EINVAL;
    if (!aligned16(q) || !aligned16(k) || !aligned16(dq) || !aligned16(dk) ||
        (n_hard > 0 && (!aligned16(table) || !aligned16(dtable))))
        return HMX_EINVAL;
    if (workspace_bytes < (grouped ? hmx_nce_bwd_grouped_workspace_bytes(m, d, n_rows, n_hard, n_inb, n_groups, period)
                                   : hmx_nce_bwd_workspace_bytes(m, d, n_rows, n_hard, n_inb)))
        return HMX_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NceParams p = make_params(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, temperature,
                              normalize, logits, nullptr, const_cast<float*>(lse), const_cast<float*>(q_inv),
                              const_cast<float*>(k_inv));
    p.grad_row_loss = grad_row_loss;
    p.dq = dq;
    p.dq_unit = const_cast<float*>(dq_unit);
    p.period = grouped ? (int)period : 0;
    const int C = (int)(1 + n_hard + n_inb);
    const unsigned g16 = (unsigned)hmx_ceil_div(m, 16), g4 = (unsigned)hmx_ceil_div(m, 4);
    void* span = hmx_profile_begin("nce_bwd", s);
    if (dq_unit != nullptr) {
        if (!aligned16(dq_unit)) return HMX_EINVAL;
        hipLaunchKernelGGL(nce_bwd_light_kernel, dim3(g4), dim3(256), 0, s, p);
    } else if (p.cpr <= 4) hipLaunchKernelGGL(nce_bwd_q_kernel<1>, dim3(g4), dim3(256), 0, s, p);
    else if (p.cpr <= 8) hipLaunchKernelGGL(nce_bwd_q_kernel<2>, dim3(g4), dim3(256), 0, s, p);
    else if (p.cpr <= 16) hipLaunchKernelGGL(nce_bwd_q_kernel<4>, dim3(g4), dim3(256), 0, s, p);
    else if (p.cpr <= 32) hipLaunchKernelGGL(nce_bwd_q_kernel<8>, dim3(g4), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(nce_bwd_q_kernel<16>, dim3(g4), dim3(256), 0, s, p);
    HMX_CHECK_LAUNCH();
    // d k: in-batch references (transposed sparse product), then the row's own positive term + normalisation
    if (hipMemsetAsync(dk, 0, (size_t)m * d * sizeof(float), s) != hipSuccess) return HMX_ELAUNCH;
    if (n_inb > 0) {
        WeightedRows src{q, logits, (int)n_inb, C, (int)(1 + n_hard)};
        const IdSource dst = p.inb32 ? IdSource(p.inb32, n_inb, p.inb_skip_self) : IdSource(p.inb_idx);
        const int rc = sorted_segment_add(src, true, dst, nullptr, 0, m * n_inb, d, dk, m, -1, workspace,
                                          workspace_bytes, s);
        if (rc != HMX_OK) return rc;
    }
    hipLaunchKernelGGL(nce_finish_k_kernel, dim3(g16), dim3(256), 0, s, p, dk);
    HMX_CHECK_LAUNCH();
    // d table: hard-negative references, then the normalisation of each touched row (dtable holds only this
    // call's sums: it must be zero on entry)
    if (n_hard > 0) {
        const uint32_t* skeys = nullptr;
        uint32_t sentinel = 0;
        int64_t n = m * n_hard;
        int rc;
        if (grouped) {
            n = n_groups * period;
            HmxCarver ws(workspace);
            float* rows = ws.take<float>((size_t)n * d);
            int64_t* ids = ws.take<int64_t>(n);
            const size_t used = hmx_align_up((size_t)n * d * sizeof(float), 256) + hmx_align_up((size_t)n * sizeof(int64_t), 256);
            hipLaunchKernelGGL(nce_group_rows_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, q, logits, m, C, (int)n_hard,
                               p.hard_ld, (int)period, 1, group_start, hard_ids, p.cpr, rows, ids);
            HMX_CHECK_LAUNCH();
            PlainRows src{rows};
            rc = sorted_segment_add(src, true, IdSource(ids), nullptr, 0, n, d, dtable, n_rows, -1,
                                    static_cast<char*>(workspace) + used, workspace_bytes - used, s, &skeys, &sentinel,
                                    dtable_flags);
        } else {
            WeightedRows src{q, logits, (int)n_hard, C, 1};
            rc = sorted_segment_add(src, true, IdSource(hard_ids), nullptr, 0, n, d, dtable, n_rows, -1, workspace, workspace_bytes,
                                    s, &skeys, &sentinel, dtable_flags);
        }
        if (rc != HMX_OK) return rc;
        if (normalize) {
            const unsigned grid = (unsigned)std::min<int64_t>(hmx_ceil_div(n, 16), 65536);
            hipLaunchKernelGGL(nce_project_table_rows_kernel, dim3(grid), dim3(256), 0, s, skeys, n, sentinel, table,
                               p.cpr, dtable);
            HMX_CHECK_LAUNCH();
        }
    }
    hmx_profile_end(span, s);
    return HMX_OK;
}

extern "C" int hmx_nce_bwd_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                               int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, const void* inb_idx,
                               int64_t n_inb, int64_t hard_ld, int64_t inb_flags, float temperature, int normalize,
                               float* logits, const float* lse,
                               const float* q_inv, const float* k_inv, const float* grad_row_loss,
                               const float* dq_unit, float* dq, float* dk, float* dtable, uint8_t* dtable_flags,
                               void* workspace, size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (hard_ld > 0 && hard_ld != n_hard) return HMX_EINVAL;     // every hard slot is read here
    return nce_bwd_impl(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, temperature,
                        normalize, logits, lse, q_inv, k_inv, grad_row_loss, dq_unit, dq, dk, dtable, dtable_flags, nullptr, 0, 0,
                        workspace, workspace_bytes, stream);
}

extern "C" int hmx_nce_bwd_grouped_f32(const float* q, const float* k, int64_t m, int64_t d, const float* table,
                                       int64_t n_rows, const int64_t* hard_ids, int64_t n_hard, const void* inb_idx,
                                       int64_t n_inb, int64_t hard_ld, int64_t inb_flags, float temperature, int normalize,
                                       float* logits, const float* lse,
                                       const float* q_inv, const float* k_inv, const float* grad_row_loss,
                                       const float* dq_unit, float* dq, float* dk, float* dtable, uint8_t* dtable_flags,
                                       const int64_t* group_start, int64_t n_groups, int64_t period, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (n_hard > 0 && !group_start) return HMX_EINVAL;
    return nce_bwd_impl(q, k, m, d, table, n_rows, hard_ids, n_hard, inb_idx, n_inb, hard_ld, inb_flags, temperature,
                        normalize, logits, lse, q_inv, k_inv, grad_row_loss, dq_unit, dq, dk, dtable, dtable_flags, group_start,
                        n_groups, period, workspace, workspace_bytes, stream);
}
