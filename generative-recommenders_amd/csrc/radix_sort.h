// Stable least-significant-digit radix sort of (uint32 key, value) entries for the sorted-run scatter-add.
//
// The keys are table row numbers (or a sentinel that sorts last), so only sort_bits(n_rows) bits are significant;
// they are cut into ceil(bits / 8) digits of equal width -- or, where that saves a whole pass (17 or 18 key bits:
// two passes instead of three), into digits of up to 9 bits with 512 counters per tile.  A pass is: per-tile digit histograms, an exclusive scan
// of every digit's row of tile counts, and a scatter that ranks the entries of a tile (4096 of them; 1024 in sorts of
// up to a million entries, which measured 10-25 % faster end to end that way) in index order (wave ballots
// match equal digits; one LDS counter per wave and digit, no atomics on the ranking path) and moves them through
// LDS so that neighbouring lanes write neighbouring addresses.  Entries with equal keys therefore keep their
// original order through every pass, which is what fixes the summation order of the scatter-add.
//
// The first pass takes its keys and values from functors, so the ids are turned into keys on the fly and no
// unsorted (key, value) arrays are ever written.  Launches of up to kSelfPrefixTiles tiles need no scan kernel:
// each tile sums the few histograms before it by itself (two launches per pass).
#pragma once
#include <cstdint>
#include <type_traits>

#include "hmx_common.h"

namespace hmx_sort {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kIpt = 16;                       // entries per thread ...
constexpr int kSmallIpt = 4;                   // ... and in sorts of at most kSmallSort entries, which would otherwise
constexpr int64_t kSmallSort = 1 << 20;        // run on too few workgroups (102,400 entries: 25 tiles on 256 CUs)
constexpr int kDigits = 256;                   // counters per tile (digits narrower than 8 bits use a prefix of them)
constexpr int kWideDigits = 512;               // ... of a pass with 9-bit digits
constexpr int kSelfPrefixTiles = 32;

struct Passes {
    int n;
    int shift[4], bits[4];
};
inline Passes plan_passes(int key_bits) {
    Passes p{};
    p.n = (key_bits + 7) / 8;
    if ((key_bits + 8) / 9 < p.n) p.n = (key_bits + 8) / 9;     // 9-bit digits only where they save a pass
    if (p.n < 1) p.n = 1;
    int at = 0;
    for (int i = 0; i < p.n; ++i) {
        const int left = key_bits - at, passes_left = p.n - i;
        p.bits[i] = (left + passes_left - 1) / passes_left;
        if (p.bits[i] < 1) p.bits[i] = 1;
        p.shift[i] = at;
        at += p.bits[i];
    }
    return p;
}

inline int ipt_for(int64_t n) { return n <= kSmallSort ? kSmallIpt : kIpt; }
inline int64_t n_tiles_for(int64_t n) { return hmx_ceil_div(n, (int64_t)kThreads * ipt_for(n)); }
// histogram rows (one per digit, n_tiles counters each) + the digit totals
inline size_t counter_bytes(int64_t n) {
    return hmx_align_up(((size_t)kWideDigits * n_tiles_for(n) + kWideDigits) * sizeof(uint32_t), 256);
}

template <class Val>
struct FromArrays {
    const uint32_t* keys;
    const Val* vals;
    __device__ __forceinline__ uint32_t key(int64_t i) const { return keys[i]; }
    __device__ __forceinline__ Val val(int64_t i) const { return vals[i]; }
};

template <class In, int ND, int IPT>
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(In in, int64_t n, int shift, uint32_t mask, int n_tiles,
                                                              uint32_t* __restrict__ hist) {
    __shared__ unsigned h[ND];
    const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
    for (int dg = tid; dg < ND; dg += kThreads) h[dg] = 0;
    __syncthreads();
    constexpr int kTile = kThreads * IPT;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    // all of a thread's keys are requested before the first is counted
    uint32_t dg[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t i = base + r * kThreads + tid;
        dg[r] = i < n ? ((in.key(i) >> shift) & mask) : 0;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t i = base + r * kThreads + tid;
        const bool valid = i < n;
        const uint32_t d = dg[r];
        // LDS atomics (integer: the counts do not depend on their order).  A wave whose lanes all hold the same digit
        // -- runs of equal ids, sentinels -- would serialise 64 ways on one counter, so that case adds once.
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        const unsigned long long bv = __ballot(valid);
        if (bv != 0 && __ballot(valid && d == d0) == bv && (bv & 1ull)) {     // wave-uniform
            if (lane == 0) atomicAdd(&h[d0], (unsigned)__popcll(bv));
        } else if (valid) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int dg = tid; dg < ND; dg += kThreads) hist[(int64_t)dg * n_tiles + blockIdx.x] = h[dg];
}

// exclusive prefix sum over the kThreads values of a workgroup; *total receives their sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sums /* [kWaves] in LDS */, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
        const uint32_t s = wave_sums[i];
        if (i < w) base += s;
        all += s;
    }
    __syncthreads();          // wave_sums may be reused by the caller
    *total = all;
    return base + incl - v;
}

// one workgroup per digit: its row of tile counts becomes the row of exclusive prefixes; totals[digit] = the sum.
// The row is walked 1024 counts at a time (a thread owns four consecutive ones) with a running base; the next
// chunk's counts are on their way during the scan of this one.
__global__ __launch_bounds__(kThreads) void radix_scan_kernel(uint32_t* __restrict__ hist, int n_tiles,
                                                              uint32_t* __restrict__ totals) {
    __shared__ uint32_t wave_sums[kWaves];
    uint32_t* row = hist + (int64_t)blockIdx.x * n_tiles;
    const int tid = threadIdx.x;
    uint32_t base = 0;
    uint32_t next[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) next[e] = 4 * tid + e < n_tiles ? row[4 * tid + e] : 0;
    for (int c0 = 0; c0 < n_tiles; c0 += 4 * kThreads) {
        uint32_t v[4];
        const int i = c0 + 4 * tid;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = next[e];
            next[e] = i + 4 * kThreads + e < n_tiles ? row[i + 4 * kThreads + e] : 0;
        }
        uint32_t total;
        uint32_t ex = base + block_exclusive_scan(v[0] + v[1] + v[2] + v[3], wave_sums, &total);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e < n_tiles) row[i + e] = ex;
            ex += v[e];
        }
        base += total;
    }
    if (tid == 0) totals[blockIdx.x] = base;
}

template <class In, class Val, int ND, int IPT>
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(In in, int64_t n, int shift, uint32_t mask, int n_tiles,
                                                                 const uint32_t* __restrict__ hist,
                                                                 const uint32_t* __restrict__ totals, int self_prefix,
                                                                 uint32_t* __restrict__ out_keys, Val* __restrict__ out_vals) {
    constexpr int DPT = ND / kThreads;          // digits per thread in the per-digit steps: thread t owns DPT t .. DPT t + DPT - 1
    constexpr int NBITS = ND == 256 ? 8 : 9;
    constexpr int kTile = kThreads * IPT;       // entries per workgroup
    __shared__ uint32_t cnt[kWaves][ND];        // per wave and digit: running count, then the wave's base inside the digit
    __shared__ uint32_t lstart[ND];             // where the digit starts in the tile's sorted order
    __shared__ uint32_t goff[ND];               // global position of the digit's first entry of this tile, minus lstart
    __shared__ uint32_t wave_sums[kWaves];
    __shared__ uint32_t skey[kTile];
    __shared__ Val sval[kTile];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
#pragma unroll
    for (int i = 0; i < kWaves; ++i)
#pragma unroll
        for (int dg = tid; dg < ND; dg += kThreads) cnt[i][dg] = 0;
    __syncthreads();

    // wave w owns entries [w * 64 * IPT, (w + 1) * 64 * IPT) of the tile, 64 consecutive ones per round
    const int64_t wbase = (int64_t)tile * kTile + (int64_t)w * 64 * IPT;
    uint32_t key[IPT], rank[IPT];
    Val val[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        if (i < n) {
            key[r] = in.key(i);
            val[r] = in.val(i);
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t d = valid ? ((key[r] >> shift) & mask) : 0;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < NBITS; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bb : ~bb;
        }
        const uint32_t before = __popcll(m & lt);
        uint32_t old = 0;
        if (valid) old = cnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) cnt[w][d] = old + (uint32_t)__popcll(m);   // the digit's lowest lane writes for all
        __builtin_amdgcn_wave_barrier();
        rank[r] = old + before;
    }
    __syncthreads();

    {   // per digit: this tile's count, the waves' bases inside it, and the digit's global position.  A thread owns
        // DPT consecutive digits, so the two exclusive scans run over per-thread sums and are refined inside the thread.
        uint32_t tile_cnt[DPT], before_tiles[DPT], digit_total[DPT];
        uint32_t my_cnt = 0, my_total = 0;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int dg = tid * DPT + u;
            uint32_t tc = 0;
#pragma unroll
            for (int i = 0; i < kWaves; ++i) {
                const uint32_t c = cnt[i][dg];
                cnt[i][dg] = tc;
                tc += c;
            }
            tile_cnt[u] = tc;
            if (self_prefix) {
                before_tiles[u] = 0;
                digit_total[u] = 0;
                for (int t = 0; t < n_tiles; ++t) {
                    const uint32_t c = hist[(int64_t)dg * n_tiles + t];
                    if (t < tile) before_tiles[u] += c;
                    digit_total[u] += c;
                }
            } else {
                before_tiles[u] = hist[(int64_t)dg * n_tiles + tile];
                digit_total[u] = totals[dg];
            }
            my_cnt += tile_cnt[u];
            my_total += digit_total[u];
        }
        uint32_t unused;
        uint32_t ls = block_exclusive_scan(my_cnt, wave_sums, &unused);
        uint32_t digit_base = block_exclusive_scan(my_total, wave_sums, &unused);
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int dg = tid * DPT + u;
            lstart[dg] = ls;
            goff[dg] = digit_base + before_tiles[u] - ls;
            ls += tile_cnt[u];
            digit_base += digit_total[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        if (wbase + r * 64 + lane < n) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t lp = lstart[d] + cnt[w][d] + rank[r];
            skey[lp] = key[r];
            sval[lp] = val[r];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)tile * kTile;
    const int n_here = (int)(left < kTile ? left : kTile);
    for (int j = tid; j < n_here; j += kThreads) {
        const uint32_t k = skey[j];
        const uint32_t pos = goff[(k >> shift) & mask] + (uint32_t)j;
        out_keys[pos] = k;
        out_vals[pos] = sval[j];
    }
}

// Sorts the n entries (first.key(i), first.val(i)) by the low key_bits bits of the key, stably.  The result lands in
// (keys_out, vals_out); (keys_tmp, vals_tmp) is the other half of the ping-pong (unused when one pass suffices).
// counters: counter_bytes(n) bytes.
template <class First, class Val>
int sort_entries(const First& first, int64_t n, int key_bits, uint32_t* keys_out, Val* vals_out, uint32_t* keys_tmp,
                 Val* vals_tmp, uint32_t* counters, hipStream_t s) {
    if (n <= 0) return HMX_OK;
    if (n >= ((int64_t)1 << 32)) return HMX_EINVAL;
    const Passes pp = plan_passes(key_bits);
    const int n_tiles = (int)n_tiles_for(n);
    uint32_t* hist = counters;
    uint32_t* totals = counters + (size_t)kWideDigits * n_tiles;
    const int self_prefix = n_tiles <= kSelfPrefixTiles ? 1 : 0;
    const uint32_t* src_keys = nullptr;
    const Val* src_vals = nullptr;
    for (int p = 0; p < pp.n; ++p) {
        const bool to_out = ((pp.n - 1 - p) % 2) == 0;     // the last pass writes the output arrays
        uint32_t* dk = to_out ? keys_out : keys_tmp;
        Val* dv = to_out ? vals_out : vals_tmp;
        const uint32_t mask = (1u << pp.bits[p]) - 1u;
        auto run_tiles = [&](const auto& in, auto nd_tag, auto ipt_tag) {
            constexpr int ND = decltype(nd_tag)::value;
            constexpr int IPT = decltype(ipt_tag)::value;
            using In = std::decay_t<decltype(in)>;
            hipLaunchKernelGGL((radix_hist_kernel<In, ND, IPT>), dim3(n_tiles), dim3(kThreads), 0, s, in, n, pp.shift[p],
                               mask, n_tiles, hist);
            if (!self_prefix) hipLaunchKernelGGL(radix_scan_kernel, dim3(ND), dim3(kThreads), 0, s, hist, n_tiles, totals);
            hipLaunchKernelGGL((radix_scatter_kernel<In, Val, ND, IPT>), dim3(n_tiles), dim3(kThreads), 0, s, in, n,
                               pp.shift[p], mask, n_tiles, hist, totals, self_prefix, dk, dv);
        };
        auto run_pass = [&](const auto& in, auto nd_tag) {
            if (ipt_for(n) == kSmallIpt) run_tiles(in, nd_tag, std::integral_constant<int, kSmallIpt>{});
            else run_tiles(in, nd_tag, std::integral_constant<int, kIpt>{});
        };
        const bool wide = pp.bits[p] > 8;
        if (p == 0) {
            if (wide) run_pass(first, std::integral_constant<int, kWideDigits>{});
            else run_pass(first, std::integral_constant<int, kDigits>{});
        } else {
            const FromArrays<Val> in{src_keys, src_vals};
            if (wide) run_pass(in, std::integral_constant<int, kWideDigits>{});
            else run_pass(in, std::integral_constant<int, kDigits>{});
        }
        src_keys = dk;
        src_vals = dv;
    }
    return HMX_OK;
}

}  // namespace
}  // namespace hmx_sort
