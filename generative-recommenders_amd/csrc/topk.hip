// Full-table inner-product scoring fused with a per-lane partial top-K (no score matrix in HBM).
//
// Geometry.  One 64-lane wave = one workgroup = (a tile of 32*NB queries) x (one item range).
// Scores are produced by 32x32 MFMA tiles with ITEMS on the rows (A operand) and QUERIES on
// the columns (B operand), so in the accumulator layout
//     col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// every lane owns ONE query and sixteen of the tile's items: the top-K filter needs no
// cross-lane traffic.  Each lane keeps, for its (query, item subset), a private candidate
// list in a workspace buffer: a score enters when it beats the lane's threshold; when the
// list is nearly full the wave sorts it cooperatively (all-pairs rank through v_readlane),
// keeps the best KP and raises the threshold to the KP-th best.  A second kernel re-scores
// every surviving candidate exactly (fp32 fma chain in k order over the fp32 table), extracts
// the K best with the (score desc, id asc) order and -- for the bf16 coarse pass -- proves
// per query that nothing dropped could have ranked: K-th exact score > largest threshold +
// error bound of the coarse score.  Queries failing the proof are flagged, not guessed.
//
// Coarse pass, bf16:  v_mfma_f32_32x32x16_bf16, item rows read straight from L2/HBM as
//   16-byte fragments (each A fragment feeds NB MFMAs), query fragments resident in VGPRs.
// Coarse pass, fp32:  v_mfma_f32_32x32x2_f32, bit-identical to the fma chain, so its
//   thresholds are exact and every query is certain.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "hmx_common.h"
#include "profile.h"

namespace {

using short8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr float kNegInf = -INFINITY;
constexpr int kRoom = 16;  // free slots guaranteed before a tile's epilogue (one lane sees 16 scores per tile)

// Candidate slots per lane list.  128 for both shortlist sizes: with KP = 32 the list is compacted every
// ~80 appends instead of every 16, which more than pays for sorting two entries per lane.
constexpr int cap_for(int KP) { return KP <= 32 ? 128 : 2 * KP; }

struct CoarseParams {
    const uint16_t* q_bf16;   // (n_q_pad, dpad) bf16   [bf16 pass]
    const uint16_t* t_bf16;   // (n_rows, dpad) bf16    [bf16 pass]
    const float* q_f32;       // (n_q, d)               [fp32 pass]
    const float* t_f32;       // (n_rows, d)            [fp32 pass]
    int64_t n_q;              // queries in this chunk
    int64_t n_rows;
    int d, dpad;
    int n_qtiles, n_splits;
    int64_t rows_per_split;
    const int64_t* exclude;   // (n_q, n_excl) global ids or null
    int n_excl;
    int64_t id_offset;
    const float* init_thr;    // (n_q_pad) starting threshold per query or null (-inf)   [bf16 pass]
    int* ring_error;          // set to 1 if a wave of the ring kernel gave up waiting (results then flagged uncertain)
    int debug_skip_publish;   // test hook: stage whose publication workgroup 0 leaves out (-1: none)
    uint2* cand;              // [wave][g][lane][CAP] (score bits, local row)
    int* meta_cnt;            // [wave][g][lane]
    float* meta_thr;          // [wave][g][lane]
    const int32_t* only_flagged;   // (n_q) or null.  Non-null: query tiles without a flagged query do nothing  [fp32 pass]
};

__device__ __forceinline__ void decode_wave(const CoarseParams& p, int& qtile, int& split) {
    const int b = blockIdx.x;
    if (p.n_splits % 8 == 0) {
        // blocks b and b+8 share an XCD (round-robin dispatch): keep a split's item range on one L2
        const int x = b & 7, r = b >> 3;
        qtile = r % p.n_qtiles;
        split = x + 8 * (r / p.n_qtiles);
    } else {
        qtile = b % p.n_qtiles;
        split = b / p.n_qtiles;
    }
}

// Cooperative compaction of one lane's candidate list (wave-uniform call).
// buf: that lane's CAP slots; n: its entry count.  Leaves min(n, KP) entries sorted by
// (score desc, arrival asc == row asc) and returns the new threshold.
template <int KP>
__device__ __forceinline__ void compact_list(uint2* buf, int n, int& new_cnt, float& new_thr, float old_thr) {
    constexpr int CAP = cap_for(KP), EPL = CAP / 64;
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends (buffer stores) have reached memory
    float s[EPL];
    unsigned id[EPL];
    int rank[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = lane + 64 * t;
        uint2 v = (e < n) ? buf[e] : make_uint2(__float_as_uint(kNegInf), 0xffffffffu);
        s[t] = __uint_as_float(v.x);
        id[t] = v.y;
        rank[t] = 0;
    }
#pragma unroll
    for (int t2 = 0; t2 < EPL; ++t2) {
        const int lim = min(64, n - 64 * t2);
        for (int f = 0; f < lim; ++f) {
            const float sf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[t2]), f));
            const int ef = f + 64 * t2;
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const int e = lane + 64 * t;
                rank[t] += (sf > s[t] || (sf == s[t] && ef < e)) ? 1 : 0;
            }
        }
    }
    new_thr = old_thr;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = lane + 64 * t;
        const bool valid = e < n;
        if (valid && rank[t] < KP) buf[rank[t]] = make_uint2(__float_as_uint(s[t]), id[t]);
        const unsigned long long m = __ballot(valid && rank[t] == KP - 1);
        if (m) {
            const int src = __ffsll((long long)m) - 1;
            new_thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[t]), src));
        }
    }
    new_cnt = n < KP ? n : KP;
}

// Per-tile filter for one group of 32 queries.  acc: the 32x32 score tile; row0: local table row of accumulator
// register 0 for this lane (tile base + 4 * (lane >> 5)).  Rows >= row_end (only possible in the last, partial
// tile of an item range) must already hold -inf in acc.
template <int KP, bool EXCL>
__device__ __forceinline__ void tile_filter(const f32x16& acc, float& thr, int& cnt, uint2* group_buf, int row0,
                                            const CoarseParams& p, int64_t q_global) {
    constexpr int CAP = cap_for(KP);
    // Maximum of the sixteen scores as a tree of v_max3 over five triples (+ register 15): the same eight
    // instructions as a chain, and the slow path can tell from the triple maxima which registers to look at.
    float t[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) t[g] = fmaxf(fmaxf(acc[3 * g], acc[3 * g + 1]), acc[3 * g + 2]);
    const float m = fmaxf(fmaxf(fmaxf(t[0], t[1]), acc[15]), fmaxf(fmaxf(t[2], t[3]), t[4]));
    if (__builtin_expect(__ballot(m > thr) == 0ull, 1)) return;  // wave-uniform fast path (falls through)
    // keep the sixteen row numbers out of the fast path: without this the compiler computes them before the branch
    asm volatile("" : "+v"(row0));
    // Appends go through a buffer descriptor built from the wave-uniform list base: scalar base + 32-bit lane
    // offset, so no per-lane 64-bit pointer has to stay live (or be spilled and reloaded) around the MFMA loop.
    const uint64_t gb = reinterpret_cast<uint64_t>(group_buf);   // wave-uniform by contract; say so to the compiler
    void* list_base = reinterpret_cast<void*>(
        ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gb >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)gb));
    const auto lists = __builtin_amdgcn_make_buffer_rsrc(list_base, 0, 64 * CAP * (int)sizeof(uint2), 0x00020000);
    // lane number recomputed here (two mbcnt) behind an opaque zero, so that nothing is carried through the hot loop
    int opaque_zero = 0;
    asm volatile("" : "+v"(opaque_zero));
    const int lane_off = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)opaque_zero)) *
                         CAP * (int)sizeof(uint2);
    auto append_row = [&](int r) {   // registers in ascending order = rows in ascending order within a lane
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        bool pass = acc[r] > thr;
        if (EXCL) {
            if (pass && q_global < p.n_q) {
                const int64_t gid = row + p.id_offset;
                const int64_t* ex = p.exclude + q_global * p.n_excl;
                for (int e = 0; e < p.n_excl; ++e) pass = pass && (ex[e] != gid);
            }
        }
        // no branch and no exec juggling per row: a lane whose row fails stores at an offset beyond the
        // descriptor's range, which the buffer range check drops
        u32x2 entry;
        entry[0] = __float_as_uint(acc[r]);
        entry[1] = (unsigned)row;
        const int off = pass ? lane_off + cnt * (int)sizeof(uint2) : (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b64(entry, lists, off, 0, 0);
        cnt += pass ? 1 : 0;
    };
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        if (__ballot(t[g] > thr) != 0ull) {   // wave-uniform: usually one triple of the five holds the hit
            append_row(3 * g);
            append_row(3 * g + 1);
            append_row(3 * g + 2);
        }
    }
    if (__ballot(acc[15] > thr) != 0ull) append_row(15);
    unsigned long long need = __ballot(cnt > CAP - kRoom);
    while (need) {
        const int o = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int n = __builtin_amdgcn_readlane(cnt, o);
        const float t_old = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr), o));
        int nc;
        float nt;
        compact_list<KP>(group_buf + o * CAP, n, nc, nt, t_old);
        if ((int)(threadIdx.x & 63) == o) {
            cnt = nc;
            thr = nt;
        }
    }
}

// -inf for the accumulator rows of a partial tile that lie beyond the item range
__device__ __forceinline__ void mask_tail_rows(f32x16& acc, int row0, int row_end) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (row0 + (r & 3) + 8 * (r >> 2) >= row_end) acc[r] = kNegInf;
}

template <int KP>
__device__ __forceinline__ void finish_group(float& thr, int& cnt, uint2* group_buf, int* meta_cnt, float* meta_thr) {
    constexpr int CAP = cap_for(KP);
    unsigned long long need = __ballot(cnt > KP);
    while (need) {
        const int o = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int n = __builtin_amdgcn_readlane(cnt, o);
        const float t_old = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr), o));
        int nc;
        float nt;
        compact_list<KP>(group_buf + (int64_t)o * CAP, n, nc, nt, t_old);
        if ((int)(threadIdx.x & 63) == o) {
            cnt = nc;
            thr = nt;
        }
    }
    meta_cnt[threadIdx.x & 63] = cnt;
    meta_thr[threadIdx.x & 63] = thr;
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 coarse pass.  (__launch_bounds__ of 2 waves per SIMD caps the kernel at 256 registers, which also makes the
// compiler keep the MFMA accumulators in VGPRs: the filter reads them every tile and AGPR accumulators would cost
// a v_accvgpr_read each.)
template <int NB, int KS, int KP>
__global__ __launch_bounds__(64, 2) void coarse_bf16_kernel(CoarseParams p) {
    constexpr int CAP = cap_for(KP);
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    int qtile, split;
    decode_wave(p, qtile, split);
    const int64_t wave = (int64_t)qtile * p.n_splits + split;
    const int64_t row_begin = (int64_t)split * p.rows_per_split;
    const int64_t row_end = min(row_begin + p.rows_per_split, p.n_rows);
    uint2* wave_buf = p.cand + wave * NB * 64 * CAP;

    short8 bq[NB][KS];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int64_t qrow = (int64_t)qtile * 32 * NB + 32 * g + j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[g][ks] = *reinterpret_cast<const short8*>(p.q_bf16 + qrow * p.dpad + 16 * ks + 8 * h);
    }
    float thr[NB];
    int cnt[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        thr[g] = p.init_thr ? p.init_thr[(int64_t)qtile * 32 * NB + 32 * g + j] : kNegInf;
        cnt[g] = 0;
    }
    const int n_rows_here = (int)max(row_end - row_begin, (int64_t)0);
    const int n_full = n_rows_here / 32;
    const uint16_t* ap = p.t_bf16 + (row_begin + j) * p.dpad + 8 * h;
    const int64_t tile_stride = 32 * (int64_t)p.dpad;

    // Three fragment buffers, rotated by a loop unrolled three times: the loads of tile t+2 are issued before the
    // MFMAs of tile t, so an L2 / HBM miss has two tile times to land.
    short8 a0[KS], a1[KS], a2[KS];
    auto load_tile = [&](short8(&dst)[KS], int t) {
        const uint16_t* src = ap + (int64_t)t * tile_stride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const short8*>(src + 16 * ks);
    };
    auto compute_tile = [&](const short8(&a)[KS], int t, bool partial) {
        f32x16 acc[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bq[g][ks], acc[g], 0, 0, 0);
        }
        const int row0 = (int)row_begin + 32 * t + 4 * h;
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if (partial) mask_tail_rows(acc[g], row0, (int)row_end);
            tile_filter<KP, false>(acc[g], thr[g], cnt[g], wave_buf + g * 64 * CAP, row0, p, 0);
        }
    };
    int t = 0;
    if (n_full >= 5) {
        // steady state: every load is unconditional (a conditional load would make the compiler's wait-count
        // analysis assume the youngest loads are the ones being waited for, i.e. no prefetch at all)
        load_tile(a0, 0);
        load_tile(a1, 1);
        for (; t + 4 < n_full; t += 3) {
            load_tile(a2, t + 2);
            compute_tile(a0, t, false);
            load_tile(a0, t + 3);
            compute_tile(a1, t + 1, false);
            load_tile(a1, t + 4);
            compute_tile(a2, t + 2, false);
        }
    }
    for (; t < n_full; ++t) {   // the last few full tiles (or a short range): plain load-then-compute
        load_tile(a2, t);
        compute_tile(a2, t, false);
    }
    if (n_rows_here % 32) {     // partial last tile: rows beyond the range re-read the last valid row, masked to -inf
        const int64_t r = min(row_begin + 32 * (int64_t)n_full + j, row_end - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            a2[ks] = *reinterpret_cast<const short8*>(p.t_bf16 + r * p.dpad + 16 * ks + 8 * h);
        compute_tile(a2, n_full, true);
    }
#pragma unroll
    for (int g = 0; g < NB; ++g)
        finish_group<KP>(thr[g], cnt[g], wave_buf + (int64_t)g * 64 * CAP, p.meta_cnt + (wave * NB + g) * 64,
                         p.meta_thr + (wave * NB + g) * 64);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 coarse pass, shared-stream form (d = 64): the eight waves of a workgroup (eight query tiles) consume ONE
// stream of item rows through a ring of kStages LDS stages of kStageTiles tiles each, so the rows cross L2 -> CU
// once per eight query tiles instead of once per tile.  No barrier in the loop: stage s is fetched by wave
// s % 8 with global -> LDS DMA while that wave is kLookahead stages earlier in its own consumption; a stage is
// published through an LDS generation counter and retired through an LDS completion counter, so the waves may
// drift up to kStages - kLookahead stages apart (the filter's slow path makes their speeds uneven).
// Progress: the wave that is furthest behind never waits -- every stage it needs was published by a wave that is
// ahead of it, and the slot it must overwrite was retired by everyone -- so the ring cannot deadlock.  Every wait
// is bounded all the same, and the bound holds for the workgroup as a whole: a wave that gives up raises all
// sixteen counters to kRingAbort, which ends every other wave's current or next wait at once and tells it (through
// the value it has just read, no extra poll) to leave the stage loop.  An aborted workgroup consumes no slot it
// did not see published, writes the lists it has (valid rows, incomplete), and sets ring_error, which marks every
// query of the launch uncertain: they are re-run through the exact kernel.  Cost of a lost wake-up: one spin limit
// (about 0.1 s) per workgroup, not one per remaining stage.
// LDS rows are 128 B; the 16-byte chunk c of row r sits at chunk position c ^ (r & 7), which spreads the eight
// rows read by neighbouring lanes over all banks without padding (the DMA needs a dense destination).
constexpr int kRingWaves = 8;
constexpr int kStages = 8;
constexpr int kLookahead = 5;
constexpr int kStageBytes = 16 * 1024;   // a stage is 128 rows of 128 B (d = 64) or 64 rows of 256 B (d = 128)
constexpr int kRingSpinLimit = 1 << 20;   // polls (s_sleep + LDS read, ~100 ns each) before a wait gives up
constexpr int kRingAbort = 1 << 30;       // counter value that means "this workgroup has given up"

template <int NB, int KP, int KS>
__global__ __launch_bounds__(64 * kRingWaves, 2) void coarse_bf16_ring_kernel(CoarseParams p) {
    static_assert(KS == 4 || KS == 8, "row of 64 or 128 bf16");
    constexpr int kRowBytes = 32 * KS;                // one bf16 row in LDS
    constexpr int kTileBytes = 32 * kRowBytes;        // one 32-row tile
    constexpr int kStageTiles = kStageBytes / kTileBytes;   // 4 (d = 64) or 2 (d = 128)
    constexpr int kCpr = kRowBytes / 16;              // 16-byte chunks per row: 8 or 16
    constexpr int kRowsPerDma = 1024 / kRowBytes;     // rows one DMA instruction (64 lanes x 16 B) covers: 8 or 4
    constexpr int kDpad = 16 * KS;
    constexpr int CAP = cap_for(KP);
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    __shared__ int ready[kStages];   // generation (stage / kStages + 1) of the data in each slot
    __shared__ int done[kStages];    // waves that finished with the slot, summed over generations
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: list pointers stay in scalar registers
    const int n_wg_tiles = p.n_qtiles / kRingWaves;
    const int wg_tile = blockIdx.x % n_wg_tiles, split = blockIdx.x / n_wg_tiles;
    const int qtile = wg_tile * kRingWaves + w;
    const int64_t wave = (int64_t)qtile * p.n_splits + split;
    const int64_t row_begin = (int64_t)split * p.rows_per_split;
    const int64_t row_end = min(row_begin + p.rows_per_split, p.n_rows);
    // the list base is wave-uniform; held in scalar registers it costs no VGPRs and is never spilled
    uint2* wave_buf;
    {
        const uint64_t wb = reinterpret_cast<uint64_t>(p.cand + wave * NB * 64 * CAP);
        wave_buf = reinterpret_cast<uint2*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(wb >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((int)wb));
    }
    if (tid < kStages) {
        ready[tid] = 0;
        done[tid] = 0;
    }

    short8 bq[NB][KS];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int64_t qrow = (int64_t)qtile * 32 * NB + 32 * g + j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[g][ks] = *reinterpret_cast<const short8*>(p.q_bf16 + qrow * p.dpad + 16 * ks + 8 * h);
    }
    float thr[NB];
    int cnt[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        thr[g] = p.init_thr ? p.init_thr[(int64_t)qtile * 32 * NB + 32 * g + j] : kNegInf;
        cnt[g] = 0;
    }
    const int n_rows_here = (int)max(row_end - row_begin, (int64_t)0);
    const int n_tiles = (n_rows_here + 31) / 32;
    const int n_stages = (n_tiles + kStageTiles - 1) / kStageTiles;

    // This wave fetches a whole stage: 16 DMA instructions of 64 lanes x 16 B; instruction i covers kRowsPerDma rows
    // of the stage and lands dense in LDS (lane order).  Lane l serves row i * kRowsPerDma + l / kCpr and writes chunk
    // position l % kCpr, so it reads logical chunk (l % kCpr) ^ (row % kCpr) -- kRowsPerDma divides kCpr, so that
    // is (l % kCpr) ^ ((i * kRowsPerDma + l / kCpr) % kCpr); for d = 64 (8 rows per DMA = kCpr) it does not depend on i
    // and the source is one pointer plus i * 1024 B per lane.
    const int lrow = lane / kCpr, lpos = lane % kCpr;
    auto fetch_stage = [&](int s, bool clamp) {
        char* dst = smem + (s % kStages) * kStageBytes;
        const int64_t base_row = row_begin + (int64_t)s * (kStageTiles * 32) + lrow;
#pragma unroll
        for (int i = 0; i < kStageBytes / 1024; ++i) {
            const int lchunk = lpos ^ ((i * kRowsPerDma + lrow) % kCpr);
            int64_t r = base_row + kRowsPerDma * i;
            if (clamp) r = min(r, row_end - 1);
            const uint16_t* src = p.t_bf16 + r * kDpad + 8 * lchunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    // The ring's ordering needs no cache maintenance: data and counters live in LDS, which serves the waves of a
    // workgroup in issue order.  What it needs is (1) the DMA's LDS writes complete before the generation counter
    // moves (vmcnt), (2) a wave's fragment reads complete before it retires the slot (lgkmcnt), and (3) the
    // compiler not moving LDS accesses across the counter operations (the "memory" clobbers).  A release fence
    // would also wait for the wave's outstanding candidate stores to be acknowledged by memory, every stage.
    auto publish = [&](int s) {   // after the stage's DMA has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every DMA of this wave has written LDS
        if (s == p.debug_skip_publish && blockIdx.x == 0) return;   // test hook (wave-uniform)
        // generations only grow, so a maximum is a store here -- except over kRingAbort, which must survive
        if (lane == 0) __hip_atomic_fetch_max(&ready[s % kStages], s / kStages + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // Returns true when the workgroup has given up (this wave's own wait ran out, or another wave's did).
    auto wait_at_least = [&](int* flag, int target) -> bool {
        int spins = 0, v;
        // wave-uniform poll: one lane's view of the counter decides for the wave (scalar compare and branch)
        while ((v = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kRingSpinLimit) {
                if (lane == 0) *p.ring_error = 1;
                if (lane < 2 * kStages)
                    __hip_atomic_store(lane < kStages ? &ready[lane] : &done[lane - kStages], kRingAbort, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                v = kRingAbort;
                break;
            }
        }
        asm volatile("" ::: "memory");
        return v >= kRingAbort;
    };
    const int n_full_stages = n_rows_here / (kStageTiles * 32);   // stages whose rows all exist
    int pending = -1;
    // what a wave does on entering stage c: its fetching duty for stage c + kLookahead, then wait for stage c
    auto enter_stage = [&](int c) -> bool {
        const int ahead = c + kLookahead;
        if (ahead < n_stages && (ahead % kRingWaves) == w) {
            if (wait_at_least(&done[ahead % kStages], kRingWaves * (ahead / kStages))) return true;   // slot retired by all eight
            fetch_stage(ahead, ahead >= n_full_stages);
            pending = ahead;
        }
        return wait_at_least(&ready[c % kStages], c / kStages + 1);
    };
    auto leave_stage = [&](int c) {
        if (pending >= 0) {
            publish(pending);
            pending = -1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the slot are complete
        if (lane == 0) __hip_atomic_fetch_add(&done[c % kStages], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // A fragments of tile tl of a stage image: row tl * 32 + j, logical chunk 2 ks + h, swizzled by row % kCpr = j % kCpr
    int a_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = j * kRowBytes + 16 * ((2 * ks + h) ^ (j % kCpr));
    auto read_tile = [&](short8(&a)[KS], const char* img, int tl) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const short8*>(img + tl * kTileBytes + a_off[ks]);
    };
    auto compute_tile = [&](const short8(&a)[KS], int t, bool may_be_partial) {
        f32x16 acc[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[g][q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bq[g][ks], acc[g], 0, 0, 0);
        }
        const int row0 = (int)row_begin + 32 * t + 4 * h;
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if (may_be_partial && (t + 1) * 32 > n_rows_here) mask_tail_rows(acc[g], row0, (int)row_end);
            tile_filter<KP, false>(acc[g], thr[g], cnt[g], wave_buf + g * 64 * CAP, row0, p, 0);
        }
    };

    __syncthreads();                      // counters cleared
    if (w < kLookahead && w < n_stages) {
        fetch_stage(w, w >= n_full_stages);
        publish(w);
    }
    // Full stages, four tiles each (two for d = 128), software-pipelined inside the wave over two accumulator sets: while the MFMAs of
    // tile t fill one set, the filter walks the other set (tile t - 1), one query group per MFMA quarter, and the
    // fragment register a quarter has just consumed is reloaded with the next tile's.  The vector work of the
    // filter then issues in the shadow of the matrix pipe instead of after it.  The idle set starts at -inf, so
    // the first step's filter passes nothing.
    short8 a0[KS];
    bool aborted = false;
    if (n_full_stages > 0) {
        f32x16 accA[NB], accB[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int q = 0; q < 16; ++q) accB[g][q] = kNegInf;
        auto step = [&](f32x16(&cur)[NB], const f32x16(&prev)[NB], const char* next_frag, int prev_t) {
            const int prev_row0 = (int)row_begin + 32 * prev_t + 4 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    if (ks == 0) {
                        f32x16 z;
#pragma unroll
                        for (int q = 0; q < 16; ++q) z[q] = 0.f;
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[ks], bq[g][ks], z, 0, 0, 0);
                    } else {
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[ks], bq[g][ks], cur[g], 0, 0, 0);
                    }
                }
                a0[ks] = *reinterpret_cast<const short8*>(next_frag + a_off[ks]);
                // quarter ks filters query group ks of the previous tile (NB <= KS groups)
                if (ks < NB)
                    tile_filter<KP, false>(prev[ks], thr[ks], cnt[ks], wave_buf + ks * 64 * CAP, prev_row0, p, 0);
            }
        };
        static_assert(NB <= KS, "at most one filter group per MFMA quarter");
        aborted = enter_stage(0);
        if (!aborted) read_tile(a0, smem, 0);
        for (int c = 0; !aborted && c < n_full_stages; ++c) {
            const char* img = smem + (c % kStages) * kStageBytes;
            const int t = c * kStageTiles;
#ifdef HMX_RING_PRIO
            if ((c ^ (w >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
            if constexpr (kStageTiles == 4) {
                step(accA, accB, img + kTileBytes, t - 1);
                step(accB, accA, img + 2 * kTileBytes, t);
                step(accA, accB, img + 3 * kTileBytes, t + 1);
            } else {
                step(accA, accB, img + kTileBytes, t - 1);
            }
            // fragments of this stage's last tile are in registers: retire the slot, take the next stage
            const char* next = img + (kStageTiles - 1) * kTileBytes;   // no next full stage: re-read this tile (unused)
            if (c + 1 < n_full_stages) {
                leave_stage(c);
                if (enter_stage(c + 1)) {     // given up: the next stage is not there to be read
                    aborted = true;
                    break;
                }
                next = smem + ((c + 1) % kStages) * kStageBytes;
            }
            step(accB, accA, next, t + kStageTiles - 2);
        }
        if (!aborted) {
            leave_stage(n_full_stages - 1);
            const int last_row0 = (int)row_begin + 32 * (n_full_stages * kStageTiles - 1) + 4 * h;
#pragma unroll
            for (int g = 0; g < NB; ++g)
                tile_filter<KP, false>(accB[g], thr[g], cnt[g], wave_buf + g * 64 * CAP, last_row0, p, 0);
        }
    }
    if (!aborted && n_full_stages < n_stages) {       // the last, short stage: plain read-then-compute
        const int c = n_full_stages;
        aborted = enter_stage(c);
        if (!aborted) {
            const char* img = smem + (c % kStages) * kStageBytes;
            const int tiles_here = n_tiles - c * kStageTiles;
            for (int tl = 0; tl < tiles_here; ++tl) {
                read_tile(a0, img, tl);
                compute_tile(a0, c * kStageTiles + tl, true);
            }
            leave_stage(c);
        }
    }
    // a wave that leaves early may still have a stage's DMA in flight: let it land before the wave ends
    if (aborted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < NB; ++g)
        finish_group<KP>(thr[g], cnt[g], wave_buf + (int64_t)g * 64 * CAP, p.meta_cnt + (wave * NB + g) * 64,
                         p.meta_thr + (wave * NB + g) * 64);
}

// ---------------------------------------------------------------------------------------------------------------
// Warm-up thresholds for launches whose waves walk the whole table (many queries, K <= 16).  A pass that starts at
// -inf spends its first thousands of tiles appending and compacting.  This kernel scores a strided sample of
// n_sample tiles with the same MFMA tiles but keeps no lists: per lane only the M largest tile maxima (M distinct
// rows of that lane's half of the table).  The two lanes of a query then hold 2 M >= K rows whose coarse score is
// >= v = min of their M-th values, and the full pass starts from a hair below v.  The value is a heuristic, not a
// bound the result depends on: the completeness proof of the re-score kernel decides, and a query whose true
// top-K does not clear the threshold by the coarse error margin is re-run exactly like any other uncertain one.
template <int NB, int KS, int M>
__global__ __launch_bounds__(64, 2) void warm_threshold_kernel(CoarseParams p, int n_sample, int tile_stride_tiles,
                                                               float* __restrict__ out_thr, int n_out, int64_t n_valid) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int qtile = blockIdx.x;
    short8 bq[NB][KS];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int64_t qrow = (int64_t)qtile * 32 * NB + 32 * g + j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[g][ks] = *reinterpret_cast<const short8*>(p.q_bf16 + qrow * p.dpad + 16 * ks + 8 * h);
    }
    float top[NB][M];
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
        for (int i = 0; i < M; ++i) top[g][i] = kNegInf;
    const uint16_t* ap = p.t_bf16 + (int64_t)j * p.dpad + 8 * h;
    const int64_t stride = 32 * (int64_t)p.dpad * tile_stride_tiles;
    short8 a0[KS], a1[KS];
    auto load_tile = [&](short8(&dst)[KS], int t) {
        const uint16_t* src = ap + (int64_t)t * stride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const short8*>(src + 16 * ks);
    };
    auto score_tile = [&](const short8(&a)[KS]) {
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bq[g][ks], acc, 0, 0, 0);
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
#pragma unroll
            for (int i = 0; i < M; ++i) {       // insert the tile maximum into the descending list
                const float hi = fmaxf(top[g][i], m);
                m = fminf(top[g][i], m);
                top[g][i] = hi;
            }
        }
    };
    // sampled tiles are full tiles by construction (the host keeps (n_sample - 1) * stride + 1 <= full tiles)
    load_tile(a0, 0);
    int t = 0;
    for (; t + 2 < n_sample; t += 2) {
        load_tile(a1, t + 1);
        score_tile(a0);
        load_tile(a0, t + 2);
        score_tile(a1);
    }
    score_tile(a0);
    if (t + 1 < n_sample) {
        load_tile(a1, t + 1);
        score_tile(a1);
    }
    if (n_out > 0) {
        // the n_out largest sampled values of each half of the sampled rows, 2 * n_out per query, for a caller that
        // combines the samples of several item ranges (rows of padded queries are not written)
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const int64_t q = (int64_t)qtile * 32 * NB + 32 * g + j;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                float v = top[g][i];
                if (v > kNegInf) v = v - fabsf(v) * 1e-6f - 1e-30f;
                if (i < n_out && q < n_valid) out_thr[(q * 2 + h) * n_out + i] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        float v = top[g][M - 1];
        v = fminf(v, __shfl_xor(v, 32));
        if (v > kNegInf) v = v - fabsf(v) * 1e-6f - 1e-30f;
        if (h == 0) out_thr[(int64_t)qtile * 32 * NB + 32 * g + j] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Start thresholds for launches of FEW queries with a LARGE k (the hard-negative miner: 1024 users, top-50, the pass
// cut into 64 item splits).  Such a pass cannot start at -inf (every score is appended, every list compacted), and
// growing exact prefix passes cost more than the pass itself.  Instead a strided sample of the table's tiles is
// scored once with the MFMA tiles of the pass and NO lists: wave (query tile, split) keeps, per lane, the maximum of
// its even and of its odd sampled tiles.  Every maximum belongs to a different row, so a query ends up with
// 4 * n_splits scores of distinct rows; group_threshold_kernel turns them into a start value that at least k
// ELIGIBLE rows reach (it knows the exact scores of the query's excluded ids).  Like every start value it is a
// heuristic the result does not depend on: the re-score kernel's completeness proof decides.
template <int KS>
__global__ __launch_bounds__(64, 2) void group_max_kernel(CoarseParams p, int tiles_per_split, int tile_step,
                                                          float* __restrict__ gmax) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    int qtile, split;
    decode_wave(p, qtile, split);
    short8 bq[KS];
    const int64_t qrow = (int64_t)qtile * 32 + j;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[ks] = *reinterpret_cast<const short8*>(p.q_bf16 + qrow * p.dpad + 16 * ks + 8 * h);
    // sampled tile i of this split is table tile (i * n_splits + split) * tile_step (all of them full tiles: host)
    const uint16_t* ap = p.t_bf16 + ((int64_t)split * tile_step * 32 + j) * p.dpad + 8 * h;
    const int64_t stride = 32 * (int64_t)p.dpad * tile_step * p.n_splits;
    float best[2] = {kNegInf, kNegInf};
    auto load_tile = [&](short8(&dst)[KS], int t) {
        const uint16_t* src = ap + (int64_t)t * stride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const short8*>(src + 16 * ks);
    };
    auto score_tile = [&](const short8(&a)[KS], float& into) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bq[ks], acc, 0, 0, 0);
        float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 3) m = fmaxf(m, fmaxf(fmaxf(acc[r], acc[r + 1]), acc[r + 2]));
        into = fmaxf(into, fmaxf(m, acc[15]));
    };
    // A ring of kDepth sampled tiles in flight (a wave is alone with its loads: with two, a tile's MFMAs -- a hundred
    // cycles -- waited a whole memory round trip for the next tile).  Even tiles feed best[0], odd ones best[1].
    constexpr int kDepth = KS <= 4 ? 6 : 4;            // an even number
    short8 ring[kDepth][KS];
#pragma unroll
    for (int s = 0; s < kDepth - 1; ++s)
        if (s < tiles_per_split) load_tile(ring[s], s);
    for (int t0 = 0; t0 < tiles_per_split; t0 += kDepth) {
#pragma unroll
        for (int s = 0; s < kDepth; ++s) {
            const int t = t0 + s;
            if (t + kDepth - 1 < tiles_per_split) load_tile(ring[(s + kDepth - 1) % kDepth], t + kDepth - 1);
            if (t < tiles_per_split) score_tile(ring[s], best[s & 1]);
        }
    }
    float* o = gmax + ((int64_t)qtile * 32 + j) * (4 * p.n_splits) + 4 * split + 2 * h;
    o[0] = best[0];
    o[1] = best[1];
}

// One wave per query: start threshold = the largest sampled maximum v such that (maxima >= v) - (excluded ids whose
// exact score, plus the coarse error margin, reaches v) >= k; -inf when there is none.  Padded queries get +inf.
__global__ __launch_bounds__(64) void group_threshold_kernel(const float* __restrict__ gmax, int n_groups,
                                                             const float* __restrict__ q_f32, const float* __restrict__ t_f32,
                                                             int64_t n_q, int64_t n_rows, int d, int k,
                                                             const int64_t* __restrict__ exclude, int n_excl, int64_t id_offset,
                                                             const float* __restrict__ max_row_norm,
                                                             float* __restrict__ init_thr) {
    constexpr int kMaxGroups = 1024, kMaxExcl = 1024;
    __shared__ float sh_g[kMaxGroups];
    __shared__ float sh_e[kMaxExcl];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    if (q >= n_q) {                 // wave-uniform
        if (lane == 0) init_thr[q] = INFINITY;
        return;
    }
    const float* qv = q_f32 + q * d;
    for (int g = lane; g < n_groups; g += 64) sh_g[g] = gmax[q * n_groups + g];
    const int ne = min(n_excl, kMaxExcl);
    for (int e = lane; e < ne; e += 64) {
        const int64_t row = exclude[q * n_excl + e] - id_offset;
        float sc = kNegInf;
        if (row >= 0 && row < n_rows) {
            const float* tv = t_f32 + row * d;
            sc = 0.f;
            if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(tv) | reinterpret_cast<uintptr_t>(qv)) & 15) == 0) {
                // the same k-ordered chain, its operands read 16 bytes at a time and eight reads ahead (a lane walks a
                // random row alone: four bytes per dependent step made this kernel 0.29 ms at 512 exclusions x d 128)
                const float4* t4 = reinterpret_cast<const float4*>(tv);
                const float4* q4 = reinterpret_cast<const float4*>(qv);
                for (int c0 = 0; c0 < d / 4; c0 += 8) {
                    float4 a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u < d / 4 ? c0 + u : d / 4 - 1;
                        a[u] = q4[c];
                        b[u] = t4[c];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (c0 + u < d / 4) {
                            sc = __builtin_fmaf(a[u].x, b[u].x, sc);
                            sc = __builtin_fmaf(a[u].y, b[u].y, sc);
                            sc = __builtin_fmaf(a[u].z, b[u].z, sc);
                            sc = __builtin_fmaf(a[u].w, b[u].w, sc);
                        }
                    }
                }
            } else {
                for (int c = 0; c < d; ++c) sc = __builtin_fmaf(qv[c], tv[c], sc);
            }
        }
        sh_e[e] = sc;
    }
    float n2 = 0.f;
    for (int c = lane; c < d; c += 64) n2 += qv[c] * qv[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
    const float eps = 0.00398f * sqrtf(n2) * max_row_norm[0];
    __syncthreads();
    float best = kNegInf;
    for (int g = lane; g < n_groups; g += 64) {
        const float v = sh_g[g];
        if (!(v > kNegInf)) continue;
        int reach = 0, gone = 0;
        for (int o = 0; o < n_groups; ++o) reach += (sh_g[o] >= v) ? 1 : 0;
        for (int e = 0; e < ne; ++e) gone += (sh_e[e] + eps >= v) ? 1 : 0;
        if (n_excl > kMaxExcl) gone += n_excl - kMaxExcl;         // ids that were not scored count as reaching v
        if (reach - gone >= k) best = fmaxf(best, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if (lane == 0) init_thr[q] = best > kNegInf ? best - fabsf(best) * 1e-6f - 1e-30f : kNegInf;
}

// fp32 MFMA: D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) per instruction, k0 on lanes 0-31, k1 on 32-63,
// so stepping k by 2 reproduces the k-ordered fma chain bit for bit.
template <int KP, bool EXCL>
__global__ __launch_bounds__(64, 2) void coarse_f32_kernel(CoarseParams p) {
    constexpr int CAP = cap_for(KP);
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    int qtile, split;
    decode_wave(p, qtile, split);
    const int64_t wave = (int64_t)qtile * p.n_splits + split;
    const int64_t row_begin = (int64_t)split * p.rows_per_split;
    const int64_t row_end = min(row_begin + p.rows_per_split, p.n_rows);
    uint2* wave_buf = p.cand + wave * 64 * CAP;
    const int64_t q_global = (int64_t)qtile * 32 + j;
    if (p.only_flagged != nullptr) {    // wave-uniform: nothing to redo in this tile (the re-score skips its queries too)
        const bool f = q_global < p.n_q && p.only_flagged[q_global] != 0;
        if (__ballot(f) == 0ull) return;
    }
    const float* qrow = p.q_f32 + min(q_global, p.n_q - 1) * p.d;
    float thr = kNegInf;
    int cnt = 0;
    const int steps = (p.d + 1) / 2;
    for (int64_t i0 = row_begin; i0 < row_end; i0 += 32) {
        const float* trow = p.t_f32 + min(i0 + j, p.n_rows - 1) * p.d;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        for (int s = 0; s < steps; ++s) {
            const int k = 2 * s + h;
            const bool in = k < p.d;
            const float a = in ? trow[k] : 0.f;
            const float b = in ? qrow[k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (i0 + 32 > row_end) mask_tail_rows(acc, (int)i0 + 4 * h, (int)row_end);
        tile_filter<KP, EXCL>(acc, thr, cnt, wave_buf, (int)i0 + 4 * h, p, q_global);
    }
    finish_group<KP>(thr, cnt, wave_buf, p.meta_cnt + wave * 64, p.meta_thr + wave * 64);
}

// ------------------------------------------------------------------ exact re-score + select
struct RescoreParams {
    const float* q_f32;   // (n_q, d) this chunk
    const float* t_f32;   // (n_rows, d)
    int64_t n_q, n_rows;
    int d;
    int NB, KP, cap, n_splits;
    const uint2* cand;
    const int* meta_cnt;
    const float* meta_thr;
    int k;
    int64_t id_offset;
    const float* max_row_norm;  // null => exact coarse pass (error bound 0)
    const int64_t* exclude;     // (n_q, n_excl) global ids removed here (null: none, or already removed upstream)
    int n_excl;
    int64_t* out_ids;           // (n_q, k)
    float* out_scores;
    int32_t* uncertain;         // (n_q) or null
    const int* ring_error;      // non-null: a set flag marks every query of the launch uncertain
    const int32_t* only_flagged;   // (n_q) or null.  Non-null: only queries with a non-zero flag are selected and written
    float* bound_out;           // (n_q) or null.  Non-null: the completeness decision is left to the caller (lists of
                                // several item ranges are merged first): bound_out[q] = the score below which rows of
                                // this range may have been dropped (largest lane threshold + coarse error margin, -inf
                                // when nothing was dropped) and uncertain[q] only reports a failed launch
};

__device__ __forceinline__ bool ranks_before(float sa, int64_t ia, float sb, int64_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}
__device__ __forceinline__ bool ranks_before(float sa, unsigned ia, float sb, unsigned ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// (four waves per SIMD = at most 128 registers: with 1024 one-query workgroups a fourth resident workgroup per CU is a
// whole round of the launch)
template <int TPQ, int CPT>
__global__ __launch_bounds__(256, 4) void rescore_select_kernel(RescoreParams p) {
    constexpr int QPB = 256 / TPQ;
    // rows are handled as 32-bit local row numbers (order by row == order by id: ids are row + id_offset)
    constexpr unsigned kNoRow = 0xffffffffu;
    constexpr int64_t kNoId = 0x7fffffffffffffffll;
    __shared__ float sh_s[2][4];     // two copies, used by alternate selection rounds: one barrier per round
    __shared__ unsigned sh_i[2][4];
    const int t = threadIdx.x % TPQ;
    const int64_t q = (int64_t)blockIdx.x * QPB + threadIdx.x / TPQ;
    const bool q_ok = q < p.n_q && (p.only_flagged == nullptr || p.only_flagged[q] != 0);
    if (p.only_flagged != nullptr && __syncthreads_or(q_ok ? 1 : 0) == 0) return;   // no flagged query in this workgroup
    const int CAP = p.cap;
    const int n_cand = p.n_splits * 2 * p.KP;
    const int qtile = (int)(q / (32 * p.NB)), g = (int)((q / 32) % p.NB), jq = (int)(q % 32);
    const float* qv = p.q_f32 + (q_ok ? q : 0) * p.d;

    float sc[CPT];
    unsigned id[CPT];
    float tmax = kNegInf;
    auto exact_score = [&](int64_t row) {
        const float* tv = p.t_f32 + row * p.d;
        float a = 0.f;
        if ((p.d & 3) == 0 && ((reinterpret_cast<uintptr_t>(tv) | reinterpret_cast<uintptr_t>(qv)) & 15) == 0) {
            // same fma chain in k order, operands fetched 16 bytes at a time
            const float4* t4 = reinterpret_cast<const float4*>(tv);
            const float4* q4 = reinterpret_cast<const float4*>(qv);
            int k = 0;
            // four 16-byte pieces of the row (and of the query) are requested together: one memory latency per 64
            // bytes instead of one per 16 (the chain itself stays sequential)
            for (; k + 4 <= p.d / 4; k += 4) {
                float4 y[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) y[u] = t4[k + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = q4[k + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a = __builtin_fmaf(x[u].x, y[u].x, a);
                    a = __builtin_fmaf(x[u].y, y[u].y, a);
                    a = __builtin_fmaf(x[u].z, y[u].z, a);
                    a = __builtin_fmaf(x[u].w, y[u].w, a);
                }
            }
            for (; k < p.d / 4; ++k) {
                const float4 x = q4[k], y = t4[k];
                a = __builtin_fmaf(x.x, y.x, a);
                a = __builtin_fmaf(x.y, y.y, a);
                a = __builtin_fmaf(x.z, y.z, a);
                a = __builtin_fmaf(x.w, y.w, a);
            }
        } else {
            for (int k = 0; k < p.d; ++k) a = __builtin_fmaf(qv[k], tv[k], a);
        }
        return a;
    };
    auto slot_of_list = [&](int list) {   // list = 2 * split + row half
        const int64_t wave = (int64_t)qtile * p.n_splits + (list >> 1);
        return (wave * p.NB + g) * 64 + jq + 32 * (list & 1);
    };
    if constexpr (TPQ == 256) {
        // One query per workgroup, up to 256 lists that are mostly short: the entries are numbered consecutively
        // (exclusive prefix of the list lengths) and dealt to the threads round-robin, so every lane re-scores about
        // the same number of rows whatever the lengths are.
        __shared__ int sh_off[257];
        __shared__ int sh_wave_total[4];
        const int n_lists = p.n_splits * 2;
        int len = 0;
        if (q_ok && t < n_lists) {
            const int64_t ls = slot_of_list(t);
            len = min(p.meta_cnt[ls], p.KP);
            tmax = p.meta_thr[ls];
        }
        int incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if ((t & 63) >= off) incl += up;
        }
        if ((t & 63) == 63) sh_wave_total[t >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (t >> 6); ++w) base += sh_wave_total[w];
        sh_off[t + 1] = base + incl;
        if (t == 0) sh_off[0] = 0;
        __syncthreads();
        const int n_valid = sh_off[256];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int jv = t + 256 * i;
            sc[i] = kNegInf;
            id[i] = kNoRow;
            if (jv < n_valid) {
                int lo = 0, hi = n_lists - 1;          // largest list with sh_off[list] <= jv
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sh_off[mid] <= jv) lo = mid; else hi = mid - 1;
                }
                const int64_t row = p.cand[slot_of_list(lo) * CAP + (jv - sh_off[lo])].y;
                sc[i] = exact_score(row);
                id[i] = (unsigned)row;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = t + TPQ * i;
            sc[i] = kNegInf;
            id[i] = kNoRow;
            if (q_ok && c < n_cand) {
                const int list = c / p.KP, e = c % p.KP;
                const int64_t lane_slot = slot_of_list(list);
                if (e == 0) tmax = fmaxf(tmax, p.meta_thr[lane_slot]);
                if (e < p.meta_cnt[lane_slot]) {
                    const int64_t row = p.cand[lane_slot * CAP + e].y;
                    sc[i] = exact_score(row);
                    id[i] = (unsigned)row;
                }
            }
        }
    }
    // largest coarse threshold of any lane that served this query
    for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off));
    if (TPQ == 256) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh_s[0][threadIdx.x >> 6] = tmax;
        __syncthreads();
        tmax = fmaxf(fmaxf(sh_s[0][0], sh_s[0][1]), fmaxf(sh_s[0][2], sh_s[0][3]));
        __syncthreads();
    }
    // Exclusion lists are applied lazily: only the row that wins a round is looked up; an excluded winner is dropped
    // and the round repeated (every repeat removes a candidate, so the loop ends).  The first 256 entries of the list
    // are staged in LDS; each wave looks the winner up by itself (four entries per lane): all waves of a query reach the same
    // verdict without a barrier.
    const int64_t* ex = (p.exclude != nullptr && q_ok) ? p.exclude + q * p.n_excl : nullptr;
    const int lane = threadIdx.x & 63;
    __shared__ int64_t sh_ex[QPB][256];
    int64_t* my_ex = sh_ex[threadIdx.x / TPQ];
    for (int x = t; x < 256; x += TPQ) my_ex[x] = (ex != nullptr && x < p.n_excl) ? ex[x] : kNoId;
    __syncthreads();
    float kth = kNegInf;
    int round = 0;
    // Each thread caches the best of its own CPT slots; a round only re-reduces the cached values, and only the wave
    // that owned the winner clears it and rescans (ids are unique within a query: the lists cover disjoint rows).
    float lbs;
    unsigned lbi;
    auto rescan = [&]() {
        lbs = sc[0];
        lbi = id[0];
#pragma unroll
        for (int i = 1; i < CPT; ++i)
            if (ranks_before(sc[i], id[i], lbs, lbi)) {
                lbs = sc[i];
                lbi = id[i];
            }
    };
    rescan();
    for (int r = 0; r < p.k; ++round) {
        float bs = lbs;
        unsigned bi = lbi;
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const unsigned oi = (unsigned)__shfl_xor((int)bi, off);
            if (ranks_before(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        if (TPQ == 256) {
            // A wave that is one round ahead writes the other copy; it cannot be two ahead (the barrier below).
            const int pb = round & 1;
            if ((threadIdx.x & 63) == 0) {
                sh_s[pb][threadIdx.x >> 6] = bs;
                sh_i[pb][threadIdx.x >> 6] = bi;
            }
            __syncthreads();
            bs = sh_s[pb][0];
            bi = sh_i[pb][0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (ranks_before(sh_s[pb][w], sh_i[pb][w], bs, bi)) {
                    bs = sh_s[pb][w];
                    bi = sh_i[pb][w];
                }
        }
        const bool found = bi != kNoRow;
        const int64_t gid = (int64_t)bi + p.id_offset;
        const bool mine = found && lbi == bi;
        if (__ballot(mine) != 0ull) {   // wave-uniform: the winner sat in this wave
            if (mine) {
#pragma unroll
                for (int i = 0; i < CPT; ++i)
                    if (id[i] == bi) {
                        sc[i] = kNegInf;
                        id[i] = kNoRow;
                    }
                rescan();
            }
        }
        int gone = 0;
        if (found && ex != nullptr) {
            gone = (my_ex[lane] == gid || my_ex[lane + 64] == gid || my_ex[lane + 128] == gid || my_ex[lane + 192] == gid) ? 1 : 0;
            for (int x = lane + 256; x < p.n_excl; x += 64) gone |= (ex[x] == gid) ? 1 : 0;
        }
        if (__ballot(gone != 0) != 0ull) continue;          // wave-local verdict, identical in every wave of the query
        if (q_ok && t == 0) {
            p.out_ids[q * p.k + r] = found ? gid : -1;
            p.out_scores[q * p.k + r] = found ? bs : kNegInf;
        }
        kth = found ? bs : kNegInf;
        ++r;
    }
    if ((p.uncertain != nullptr || p.bound_out != nullptr) && q_ok && t == 0) {
        int flag = 0;
        float bound = kNegInf;
        if (tmax > kNegInf) {
            float eps = 0.f;
            if (p.max_row_norm != nullptr) {
                float n2 = 0.f;
                for (int k = 0; k < p.d; ++k) n2 = __builtin_fmaf(qv[k], qv[k], n2);
                // |bf16 coarse score - exact score| <= (2^-8 + small) * |q| * max|t|  (two bf16 roundings per product)
                eps = 0.00398f * sqrtf(n2) * p.max_row_norm[0];
            }
            bound = tmax + eps;
            flag = !(kth > bound);
        }
        if (p.bound_out != nullptr) {
            p.bound_out[q] = bound;
            flag = 0;
        }
        if (p.ring_error != nullptr && *p.ring_error != 0) flag = 1;
        if (p.uncertain != nullptr) p.uncertain[q] = flag;
    }
}

// ------------------------------------------------------------------ conversions / merge
__global__ __launch_bounds__(256) void to_bf16_padded_kernel(const float* __restrict__ src, int64_t rows, int d,
                                                             int64_t rows_pad, int dpad, uint16_t* __restrict__ dst,
                                                             unsigned* __restrict__ max_norm_bits) {
    // one group of 16 lanes per output row, 4 consecutive elements per lane and step (dpad is a multiple of 16);
    // the row norm is reduced inside the group and the block's largest norm goes out through one atomic
    __shared__ float wave_max[4];
    const int c = threadIdx.x & 15;
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    float best = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < (rows_pad + 15) / 16 * 16;
         row += (int64_t)gridDim.x * 16) {
        float n2 = 0.f;
        if (row < rows_pad) {
            for (int k = 4 * c; k < dpad; k += 64) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows) {
                    if (vec && k + 4 <= d) {
                        v = *reinterpret_cast<const float4*>(src + row * d + k);
                    } else {
                        if (k < d) v.x = src[row * d + k];
                        if (k + 1 < d) v.y = src[row * d + k + 1];
                        if (k + 2 < d) v.z = src[row * d + k + 2];
                        if (k + 3 < d) v.w = src[row * d + k + 3];
                    }
                }
                n2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                uint2 o;
                o.x = (uint32_t)hmx_f32_to_bf16(v.x) | ((uint32_t)hmx_f32_to_bf16(v.y) << 16);
                o.y = (uint32_t)hmx_f32_to_bf16(v.z) | ((uint32_t)hmx_f32_to_bf16(v.w) << 16);
                *reinterpret_cast<uint2*>(dst + row * dpad + k) = o;
            }
        }
        for (int off = 8; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
        best = fmaxf(best, n2);
    }
    if (max_norm_bits != nullptr) {
        for (int off = 32; off >= 16; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = best;
        __syncthreads();
        // slightly inflated so that the rounding of this reduction can never under-state a norm
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
            atomicMax(max_norm_bits, __float_as_uint(sqrtf(m) * 1.0001f));
        }
    }
}

__global__ __launch_bounds__(64) void topk_merge_kernel(const int64_t* __restrict__ cand_ids,
                                                        const float* __restrict__ cand_scores, int n_shards,
                                                        int64_t n_q, int k, int64_t* __restrict__ out_ids,
                                                        float* __restrict__ out_scores) {
    constexpr int64_t kNoId = 0x7fffffffffffffffll;
    constexpr int CPT = 16;  // up to 1024 candidates (n_shards * k)
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const int n_cand = n_shards * k;
    float sc[CPT];
    int64_t id[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = lane + 64 * i;
        sc[i] = kNegInf;
        id[i] = kNoId;
        if (c < n_cand) {
            const int s = c / k, e = c % k;
            const int64_t ci = cand_ids[((int64_t)s * n_q + q) * k + e];
            if (ci >= 0) {
                id[i] = ci;
                sc[i] = cand_scores[((int64_t)s * n_q + q) * k + e];
            }
        }
    }
    for (int r = 0; r < k; ++r) {
        float bs = sc[0];
        int64_t bi = id[0];
#pragma unroll
        for (int i = 1; i < CPT; ++i)
            if (ranks_before(sc[i], id[i], bs, bi)) {
                bs = sc[i];
                bi = id[i];
            }
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int64_t oi = __shfl_xor(bi, off);
            if (ranks_before(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        const bool found = bi != kNoId;
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (found && id[i] == bi) {
                sc[i] = kNegInf;
                id[i] = kNoId;
            }
        if (lane == 0) {
            out_ids[q * k + r] = found ? bi : -1;
            out_scores[q * k + r] = found ? bs : kNegInf;
        }
    }
}

// k <= 16: one THREAD per query keeps the best k (score, id) pairs seen so far in a sorted register array and inserts
// every candidate (one compare-exchange per slot).  A wave's loads of one list position sit 8 k bytes apart, so the k
// positions of a list re-use the same cache lines; 8 lists x 262,144 queries x 10: 0.84 ms with a wave per query.
__global__ __launch_bounds__(256) void topk_merge_insert_kernel(const int64_t* __restrict__ cand_ids,
                                                                const float* __restrict__ cand_scores, int n_shards,
                                                                int64_t n_q, int k, int64_t* __restrict__ out_ids,
                                                                float* __restrict__ out_scores) {
    constexpr int64_t kNoId = 0x7fffffffffffffffll;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_q) return;
    float ts[16];
    int64_t ti[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        ts[i] = kNegInf;
        ti[i] = kNoId;
    }
    for (int s = 0; s < n_shards; ++s) {
        const int64_t base = ((int64_t)s * n_q + q) * k;
        for (int e = 0; e < k; ++e) {
            int64_t ci = cand_ids[base + e];
            if (ci < 0) continue;
            float cs = cand_scores[base + e];
#pragma unroll
            for (int i = 0; i < 16; ++i) {            // the better of (slot, carried) stays, the other moves on
                const bool up = ranks_before(cs, ci, ts[i], ti[i]);
                const float s2 = up ? ts[i] : cs;
                const int64_t i2 = up ? ti[i] : ci;
                ts[i] = up ? cs : ts[i];
                ti[i] = up ? ci : ti[i];
                cs = s2;
                ci = i2;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < k) {
            const bool found = ti[i] != kNoId;
            out_ids[q * k + i] = found ? ti[i] : -1;
            out_scores[q * k + i] = found ? ts[i] : kNegInf;
        }
}

// Starting thresholds for the full pass from the exact top-K of a sample of the rows (scores (n_q, k), best
// first).  K rows of the sample score at least t = scores[q][k-1] exactly, so the K-th best exact score over ALL
// rows is >= t, and every row of the final top-K has a coarse score >= t - eps (eps = bound of the bf16 coarse
// error).  The threshold sits a hair below t - eps so that the completeness proof of the re-score kernel
// (K-th exact score > largest threshold + eps) still holds when no list ever raised its threshold further.
// Queries beyond n_q (padding of the last tile) get +inf: they never append.
__global__ __launch_bounds__(256) void seed_threshold_kernel(const float* __restrict__ scores, const float* __restrict__ q,
                                                             int64_t n_q, int64_t n_q_pad, int d, int k,
                                                             const float* __restrict__ max_row_norm,
                                                             float* __restrict__ init_thr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_q_pad) return;
    if (i >= n_q) {
        init_thr[i] = INFINITY;
        return;
    }
    const float t = scores[i * k + k - 1];
    if (!(t > kNegInf)) {           // fewer than k sample rows survived the exclusions
        init_thr[i] = kNegInf;
        return;
    }
    float n2 = 0.f;
    for (int c = 0; c < d; ++c) n2 = __builtin_fmaf(q[i * d + c], q[i * d + c], n2);
    const float eps = 0.00398f * sqrtf(n2) * max_row_norm[0];
    const float lo = t - eps;
    init_thr[i] = lo - fmaxf(fabsf(lo), eps) * 1e-5f - 1e-30f;
}

// ------------------------------------------------------------------ host-side planning
struct Plan {
    int NB, KP, KS, dpad;
    int64_t chunk;       // queries per launch
    int n_qtiles;        // per full chunk
    int n_splits;
    int64_t rows_per_split;
    size_t q_bf16_bytes, cand_bytes, meta_bytes, thr_bytes;
    size_t total() const { return q_bf16_bytes + cand_bytes + 2 * meta_bytes + thr_bytes + 1280; }
};

// out[q] = k-th largest of the n_slabs * n_values numbers vals[s * slab_stride + q * n_values + v] (-inf when there are
// fewer than k).  One thread per query keeps the k largest seen so far in a sorted register array (k <= 16).
__global__ __launch_bounds__(256) void kth_largest_kernel(const float* __restrict__ vals, int n_slabs, int64_t slab_stride,
                                                          int64_t n_q, int n_values, int k, float* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_q) return;
    float top[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) top[i] = kNegInf;
    for (int sl = 0; sl < n_slabs; ++sl) {
        const float* src = vals + sl * slab_stride + q * n_values;
        for (int v = 0; v < n_values; ++v) {
            float m = src[v];
#pragma unroll
            for (int i = 0; i < 16; ++i) {      // insert into the descending list
                const float hi = fmaxf(top[i], m);
                m = fminf(top[i], m);
                top[i] = hi;
            }
        }
    }
    float r = kNegInf;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i == k - 1) r = top[i];
    out[q] = r;
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ dst, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v;
}

int dpad_for(int64_t d) {
    int p = 16;
    while (p < d) p *= 2;
    return p;
}

// The shared-stream kernel exists for rows of 64 and of 128 bf16 (four or eight 16-wide K steps) with up to
// 16 / KS query groups per wave.
bool ring_enabled() {
    static const bool on = [] {
        const char* e = getenv("HMX_TOPK_RING");
        return e == nullptr || e[0] != '0';
    }();
    return on;
}
// With fewer than four query groups per wave (few queries) a workgroup's eight waves cover only 256 or 512 queries,
// so the shared stream needs at least 512 queries to fill its waves, and a pass over a short table prefix (the
// seeding passes) is left to the independent-wave kernel.
constexpr int64_t kRingMinRowsFewQueries = 1 << 18;
bool ring_shape(bool bf16, int NB, int KS, int64_t n_q) {
    return bf16 && (KS == 4 || KS == 8) && ring_enabled() && (NB == 4 || (KS == 8 && NB == 2) || n_q >= 512);
}
bool ring_pass(int NB, int KS, int64_t rows) { return NB == 4 || (KS == 8 && NB == 2) || rows >= kRingMinRowsFewQueries; }

int g_debug_skip_publish = -1;   // see hmx_debug_ring_skip_publish

bool make_plan(int64_t n_q, int64_t n_rows, int64_t d, int64_t k, bool bf16, Plan& pl) {
    if (n_q <= 0 || n_rows <= 0 || d <= 0 || k <= 0 || k > 64 || n_rows >= ((int64_t)1 << 31)) return false;
    if (bf16 && d > 128) return false;
    pl.KP = (k <= 16) ? 32 : 64;
    pl.dpad = dpad_for(d);
    pl.KS = pl.dpad / 16;
    pl.NB = 1;
    if (bf16) pl.NB = n_q >= 16384 ? 4 : (n_q >= 4096 ? 2 : 1);
    if (bf16 && pl.KS == 8 && pl.NB == 4) pl.NB = 2;   // d = 128: four query groups would not fit 256 registers
    const int64_t qpw = 32 * pl.NB;   // queries per wave
    pl.chunk = std::min<int64_t>(hmx_ceil_div(n_q, qpw) * qpw, 2048 * qpw);
    if (ring_shape(bf16, pl.NB, pl.KS, n_q)) pl.chunk = hmx_ceil_div(pl.chunk, qpw * kRingWaves) * qpw * kRingWaves;
    pl.n_qtiles = (int)(pl.chunk / (32 * pl.NB));
    // Item splits fill the chip's 2048 wave slots (256 CUs x 8) without exceeding them: one wave more than a full
    // round costs a second round, and a launch of more than half the slots' query tiles stays at one split (a
    // 213,568-query tail of a sharded search took 50.8 ms in two cold splits against 12.5 ms for a full
    // 262,144-query launch of the same 500 K-row range).
    int64_t s = std::max<int64_t>(1, 2048 / pl.n_qtiles);
    s = std::min<int64_t>(s, 8192 / (2 * pl.KP));              // candidates per query <= 8192
    s = std::min<int64_t>(s, std::max<int64_t>(1, n_rows / 256));     // at least eight tiles per wave
    if (s >= 8) s = s / 8 * 8;                                  // XCD-aware block mapping wants a multiple of 8
    pl.n_splits = (int)std::max<int64_t>(1, s);
    pl.rows_per_split = hmx_ceil_div(hmx_ceil_div(n_rows, pl.n_splits), 32) * 32;
    const int64_t waves = (int64_t)pl.n_qtiles * pl.n_splits;
    pl.q_bf16_bytes = bf16 ? hmx_align_up((size_t)pl.chunk * pl.dpad * 2, 256) : 0;
    pl.cand_bytes = hmx_align_up((size_t)waves * pl.NB * 64 * cap_for(pl.KP) * sizeof(uint2), 256);
    pl.meta_bytes = hmx_align_up((size_t)waves * pl.NB * 64 * 4, 256);
    pl.thr_bytes = bf16 ? hmx_align_up((size_t)pl.chunk * sizeof(float), 256) : 0;
    return true;
}

// The bf16 coarse pass never looks at the exclusion lists: excluded ids are removed when the shortlist is
// re-scored, and the completeness proof there covers what they displaced.
template <int NB, int KS, int KP>
void launch_bf16(const CoarseParams& p, hipStream_t s) {
    if constexpr ((KS == 4 || KS == 8) && NB * KS <= 16) {   // query fragments of a wave fit 64 registers
        if (ring_enabled() && p.n_qtiles % kRingWaves == 0 && p.ring_error != nullptr) {
            constexpr int lds = kStages * kStageBytes;
            hmx_allow_dynamic_lds(reinterpret_cast<const void*>(&coarse_bf16_ring_kernel<NB, KP, KS>), lds);
            hipLaunchKernelGGL((coarse_bf16_ring_kernel<NB, KP, KS>), dim3((p.n_qtiles / kRingWaves) * p.n_splits),
                               dim3(64 * kRingWaves), lds, s, p);
            return;
        }
    }
    hipLaunchKernelGGL((coarse_bf16_kernel<NB, KS, KP>), dim3(p.n_qtiles * p.n_splits), dim3(64), 0, s, p);
}
template <int NB, int KS>
void launch_bf16_kp(const CoarseParams& p, int KP, hipStream_t s) {
    if (KP == 32) launch_bf16<NB, KS, 32>(p, s);
    else launch_bf16<NB, KS, 64>(p, s);
}
template <int NB>
void launch_bf16_ks(const CoarseParams& p, int KS, int KP, hipStream_t s) {
    switch (KS) {
        case 1: launch_bf16_kp<NB, 1>(p, KP, s); break;
        case 2: launch_bf16_kp<NB, 2>(p, KP, s); break;
        case 4: launch_bf16_kp<NB, 4>(p, KP, s); break;
        default: launch_bf16_kp<NB, 8>(p, KP, s); break;
    }
}

// combined-sample multiplier of the threshold exchange (HMX_TOPK_SAMPLE_BOOST; measured on 8 ranges of 500 K rows with
// the k-th-of-the-union combine: 1 -> 0.6 ms of sampling + 13.3 ms pass, 2 -> 1.1 + 12.6, 4 -> 2.2 + 12.2)
int64_t sample_boost() {
    static const int64_t v = [] {
        const char* e = getenv("HMX_TOPK_SAMPLE_BOOST");
        const int64_t x = e ? atoll(e) : 1;
        return x >= 1 && x <= 64 ? x : 1;
    }();
    return v;
}

bool warm_enabled() {
    static const bool on = [] {
        const char* e = getenv("HMX_TOPK_WARM");
        return e == nullptr || e[0] != '0';
    }();
    return on;
}

template <int NB, int KS>
void launch_warm_m(int M, const CoarseParams& p, int n_qtiles, int n_sample, int stride_tiles, float* out, hipStream_t s,
                   int n_out, int64_t n_valid) {
    if (M <= 5)
        hipLaunchKernelGGL((warm_threshold_kernel<NB, KS, 5>), dim3(n_qtiles), dim3(64), 0, s, p, n_sample, stride_tiles, out,
                           n_out, n_valid);
    else
        hipLaunchKernelGGL((warm_threshold_kernel<NB, KS, 8>), dim3(n_qtiles), dim3(64), 0, s, p, n_sample, stride_tiles, out,
                           n_out, n_valid);
}
template <int NB>
void launch_warm_ks(int KS, int M, const CoarseParams& p, int n_qtiles, int n_sample, int stride_tiles, float* out,
                    hipStream_t s, int n_out, int64_t n_valid) {
    switch (KS) {
        case 1: launch_warm_m<NB, 1>(M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid); break;
        case 2: launch_warm_m<NB, 2>(M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid); break;
        case 4: launch_warm_m<NB, 4>(M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid); break;
        default: launch_warm_m<NB, 8>(M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid); break;
    }
}
void launch_warm(int NB, int KS, int M, const CoarseParams& p, int n_qtiles, int n_sample, int stride_tiles, float* out,
                 hipStream_t s, int n_out = 0, int64_t n_valid = 0) {
    if (NB == 4) launch_warm_ks<4>(KS, M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid);
    else if (NB == 2) launch_warm_ks<2>(KS, M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid);
    else launch_warm_ks<1>(KS, M, p, n_qtiles, n_sample, stride_tiles, out, s, n_out, n_valid);
}

void launch_rescore(const RescoreParams& r, hipStream_t s) {
    const int n_cand = r.n_splits * 2 * r.KP;
#define HMX_RS(TPQ, CPT)                                                                                          \
    hipLaunchKernelGGL((rescore_select_kernel<TPQ, CPT>), dim3((unsigned)hmx_ceil_div(r.n_q, 256 / TPQ)), dim3(256), 0, \
                       s, r)
    if (n_cand <= 64) HMX_RS(64, 1);
    else if (n_cand <= 128) HMX_RS(64, 2);
    else if (n_cand <= 256) HMX_RS(64, 4);
    else if (n_cand <= 1024) HMX_RS(256, 4);
    else if (n_cand <= 4096) HMX_RS(256, 16);
    else HMX_RS(256, 32);
#undef HMX_RS
}

// Threshold seeding for the bf16 pass.  A pass that starts with thresholds at -inf spends its first tiles
// appending every score and compacting every list; so growing table prefixes are scored first, each seeded with
// the exact top-K of the one before.  The first prefix is 2 * KP rows: each of a query's two lanes then sees exactly
// KP rows, which fit its list without a single compaction (a cold pass over 256 rows at KP = 64 took 0.5 ms, all of
// it in the end-of-pass compactions; over 128 rows it takes 8 us).  Every later pass sees about 16 K scores above
// its starting thresholds per query.
constexpr int kMaxSeedLevels = 4;
int seed_levels(int64_t n_rows, int KP, int64_t* sample_rows) {
    int n = 0;
    for (int64_t r = 2 * KP; n < kMaxSeedLevels && r <= 524288 && r * 4 <= n_rows; r *= 16) sample_rows[n++] = r;
    return n;
}

// Every how-many-th tile of the table group_max_kernel samples (HMX_TOPK_GROUP_STEP overrides; 0 = keep the exact
// prefix passes).
int group_max_step() {
    static const int step = [] {
        const char* e = getenv("HMX_TOPK_GROUP_STEP");
        if (e == nullptr) return 16;
        const int v = atoi(e);
        return (v >= 0 && v <= 1024) ? v : 16;
    }();
    return step;
}

// Launch shapes whose start thresholds are sampled (warm_threshold_kernel) or taken from the caller: top-16 or less
// in at most four item splits, i.e. launches of 512 query tiles or more.  Every split starts from the query's one
// sampled threshold.  Whole table, top-10, ms per launch seeded -> sampled: 131,072 queries (2 splits) 61.2 -> 54.5,
// 100,000 (2) 52.3 -> 48.8, 65,536 (4) 30.3 -> 29.1; at 5 to 7 splits the two are equal and the prefixes stay.
constexpr int kSampleMaxSplits = 4;
bool sampled_start(const Plan& pl) { return pl.KP == 32 && pl.n_splits <= kSampleMaxSplits && warm_enabled(); }

struct TopkPlans {
    Plan full;
    Plan seed[kMaxSeedLevels];
    int64_t seed_rows[kMaxSeedLevels];
    int n_seed;
    bool want_seed;      // a split pass without sampled thresholds: it is seeded (group maxima, or the prefix passes)
    size_t cand_bytes, meta_bytes;
    size_t total() const { return full.q_bf16_bytes + full.thr_bytes + cand_bytes + 2 * meta_bytes + 2048; }
};

bool make_plans(int64_t n_q, int64_t n_rows, int64_t d, int64_t k, bool bf16, TopkPlans& tp) {
    if (!make_plan(n_q, n_rows, d, k, bf16, tp.full)) return false;
    tp.n_seed = 0;
    tp.want_seed = false;
    tp.cand_bytes = tp.full.cand_bytes;
    tp.meta_bytes = tp.full.meta_bytes;
    if (!bf16) return true;
    // Seeding pays whenever the full pass is cut into item splits (launches of up to half the chip's wave slots in
    // query tiles): such a pass has no sampled start, and its cold start costs more than the prefix passes
    // (whole table, top-10, ms per launch cold -> seeded: 40,000 queries 39.8 -> 19.0, 65,536 49.4 -> 29.5,
    // 100,000 69.0 -> 50.3, 131,072 73.9 -> 59.0).  One-split launches start from sampled thresholds instead.
    int64_t rows[kMaxSeedLevels];
    static const bool seed_on = [] {
        const char* e = getenv("HMX_TOPK_SEED");
        return e == nullptr || e[0] != '0';
    }();
    const bool sampled = sampled_start(tp.full);
    tp.want_seed = tp.full.n_splits >= 2 && !sampled && seed_on;
    const int n = tp.want_seed ? seed_levels(n_rows, tp.full.KP, rows) : 0;
    for (int i = 0; i < n; ++i) {
        Plan ps;
        if (!make_plan(n_q, rows[i], d, k, true, ps) || ps.chunk != tp.full.chunk) {
            tp.n_seed = 0;
            tp.cand_bytes = tp.full.cand_bytes;
            tp.meta_bytes = tp.full.meta_bytes;
            return true;
        }
        tp.seed[i] = ps;
        tp.seed_rows[i] = rows[i];
        tp.n_seed = i + 1;
        tp.cand_bytes = std::max(tp.cand_bytes, ps.cand_bytes);
        tp.meta_bytes = std::max(tp.meta_bytes, ps.meta_bytes);
    }
    return true;
}

int run_topk(bool bf16, const float* queries, int64_t n_q, const float* table, const uint16_t* table_bf16,
             const float* max_row_norm, int64_t n_rows, int64_t d, int64_t k, const int64_t* exclude, int64_t n_excl,
             int64_t id_offset, int64_t* out_ids, float* out_scores, int32_t* uncertain, void* workspace,
             size_t workspace_bytes, hipStream_t s, const float* ext_thr = nullptr, float* bound_out = nullptr,
             float* sample_out = nullptr, int64_t sample_ranges = 1, int sample_values = 1,
             const int32_t* only_flagged = nullptr) {
    // ext_thr (n_q): start thresholds supplied by the caller (used in place of the sampled ones); bound_out (n_q):
    // see RescoreParams; sample_out (n_q, sample_values): only the sampled start thresholds are computed and written
    // (-inf where this launch shape does not sample), no scoring pass runs.  sample_values = 1: the threshold itself;
    // an even number 2 m: the m largest sampled values of each half of the sampled rows.
    if (n_q == 0) return HMX_OK;
    TopkPlans tp;
    if (!make_plans(n_q, n_rows, d, k, bf16, tp)) return HMX_EINVAL;
    const Plan& pl = tp.full;
    if (!queries || !workspace) return HMX_EINVAL;
    if (sample_out == nullptr && (!table || !out_ids || !out_scores)) return HMX_EINVAL;
    if (bf16 && (!table_bf16 || (sample_out == nullptr && (!max_row_norm || !uncertain)))) return HMX_EINVAL;
    if (exclude == nullptr) n_excl = 0;
    if (workspace_bytes < tp.total()) return HMX_EWORKSPACE;
    HmxCarver ws(workspace);
    uint16_t* q_bf16 = bf16 ? ws.take<uint16_t>((size_t)pl.chunk * pl.dpad) : nullptr;
    float* init_thr = bf16 ? ws.take<float>((size_t)pl.chunk) : nullptr;
    uint2* cand = ws.take<uint2>(tp.cand_bytes / sizeof(uint2));
    int* meta_cnt = ws.take<int>(tp.meta_bytes / 4);
    float* meta_thr = ws.take<float>(tp.meta_bytes / 4);
    int* ring_error = ws.take<int>(64);
    const bool ring = ring_shape(bf16, pl.NB, pl.KS, n_q);
    if (ring && hipMemsetAsync(ring_error, 0, sizeof(int), s) != hipSuccess) return HMX_ELAUNCH;
    const int64_t qpw = 32 * pl.NB;
    const bool excl = exclude != nullptr && n_excl > 0;

    for (int64_t q0 = 0; q0 < n_q; q0 += pl.chunk) {
        const int64_t nq = std::min(pl.chunk, n_q - q0);
        int n_qtiles = (int)hmx_ceil_div(nq, qpw);
        if (ring) n_qtiles = (int)hmx_ceil_div(n_qtiles, kRingWaves) * kRingWaves;   // padded query tiles score zeros
        const int64_t rows_pad = (int64_t)n_qtiles * qpw;
        if (bf16)
            hipLaunchKernelGGL(to_bf16_padded_kernel, dim3((unsigned)std::min<int64_t>(hmx_ceil_div(rows_pad, 16), 8192)),
                               dim3(256), 0, s, queries + q0 * d, nq, (int)d, rows_pad, pl.dpad, q_bf16,
                               (unsigned*)nullptr);
        // one coarse pass + exact re-score over the first `rows` rows of the table
        auto pass = [&](const Plan& plan, int64_t rows, const float* thr0, int32_t* flags, const char* tag) -> int {
            const int waves = n_qtiles * plan.n_splits;
            CoarseParams p{};
            p.q_bf16 = q_bf16;
            p.t_bf16 = table_bf16;
            p.q_f32 = queries + q0 * d;
            p.t_f32 = table;
            p.n_q = nq;
            p.n_rows = rows;
            p.d = (int)d;
            p.dpad = plan.dpad;
            p.n_qtiles = n_qtiles;
            p.n_splits = plan.n_splits;
            p.rows_per_split = plan.rows_per_split;
            p.exclude = exclude ? exclude + q0 * n_excl : nullptr;
            p.n_excl = (int)n_excl;
            p.id_offset = id_offset;
            p.init_thr = thr0;
            p.ring_error = (ring && ring_pass(plan.NB, plan.KS, rows)) ? ring_error : nullptr;
            p.debug_skip_publish = -1;
            if (p.ring_error != nullptr && rows == n_rows && g_debug_skip_publish >= 0) {   // one-shot test hook, full pass
                p.debug_skip_publish = g_debug_skip_publish;
                g_debug_skip_publish = -1;
            }
            p.cand = cand;
            p.meta_cnt = meta_cnt;
            p.meta_thr = meta_thr;
            p.only_flagged = (!bf16 && only_flagged != nullptr) ? only_flagged + q0 : nullptr;
            void* span = hmx_profile_begin(tag, s);
            if (bf16) {
                if (plan.NB == 4) launch_bf16_ks<4>(p, plan.KS, plan.KP, s);
                else if (plan.NB == 2) launch_bf16_ks<2>(p, plan.KS, plan.KP, s);
                else launch_bf16_ks<1>(p, plan.KS, plan.KP, s);
            } else if (plan.KP == 32) {
                if (excl) hipLaunchKernelGGL((coarse_f32_kernel<32, true>), dim3(waves), dim3(64), 0, s, p);
                else hipLaunchKernelGGL((coarse_f32_kernel<32, false>), dim3(waves), dim3(64), 0, s, p);
            } else {
                if (excl) hipLaunchKernelGGL((coarse_f32_kernel<64, true>), dim3(waves), dim3(64), 0, s, p);
                else hipLaunchKernelGGL((coarse_f32_kernel<64, false>), dim3(waves), dim3(64), 0, s, p);
            }
            hmx_profile_end(span, s);
            HMX_CHECK_LAUNCH();
            RescoreParams r{};
            r.q_f32 = queries + q0 * d;
            r.t_f32 = table;
            r.n_q = nq;
            r.n_rows = rows;
            r.d = (int)d;
            r.NB = plan.NB;
            r.KP = plan.KP;
            r.cap = cap_for(plan.KP);
            r.n_splits = plan.n_splits;
            r.cand = cand;
            r.meta_cnt = meta_cnt;
            r.meta_thr = meta_thr;
            r.k = (int)k;
            r.id_offset = id_offset;
            r.max_row_norm = bf16 ? max_row_norm : nullptr;
            // bf16 pass: exclusions are applied here.  fp32 pass: the coarse kernel already skipped them.
            r.exclude = (bf16 && excl) ? exclude + q0 * n_excl : nullptr;
            r.n_excl = (int)n_excl;
            r.out_ids = out_ids + q0 * k;
            r.out_scores = out_scores + q0 * k;
            r.uncertain = flags;
            r.only_flagged = (!bf16 && only_flagged != nullptr) ? only_flagged + q0 : nullptr;
            r.bound_out = (bound_out != nullptr && rows == n_rows) ? bound_out + q0 : nullptr;
            r.ring_error = (ring && ring_pass(plan.NB, plan.KS, rows)) ? ring_error : nullptr;
            launch_rescore(r, s);
            HMX_CHECK_LAUNCH();
            return HMX_OK;
        };
        const float* thr0 = nullptr;
        const bool samples = bf16 && tp.n_seed == 0 && sampled_start(pl);
        if (sample_out != nullptr && !samples) {
            hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)hmx_ceil_div(nq * sample_values, 256)), dim3(256), 0, s,
                               sample_out + q0 * sample_values, nq * sample_values, kNegInf);
            HMX_CHECK_LAUNCH();
            continue;
        }
        // few queries, large k, split pass: one sampled no-list pass instead of the exact prefix passes
        const int64_t gm_tiles = group_max_step() > 0 ? n_rows / 32 / (int64_t)group_max_step() : 0;
        const int gm_tps = (int)std::min<int64_t>(gm_tiles / pl.n_splits, 1 << 20);
        const bool group_seed = bf16 && tp.want_seed && pl.NB == 1 && pl.KP == 64 && group_max_step() > 0 && gm_tps >= 4 &&
                                4 * pl.n_splits <= 1024 && (size_t)rows_pad * 4 * pl.n_splits * sizeof(float) <= tp.cand_bytes &&
                                sample_out == nullptr && ext_thr == nullptr;
        if (group_seed) {
            float* gmax = reinterpret_cast<float*>(cand);     // consumed before the pass below writes its lists there
            CoarseParams gp{};
            gp.q_bf16 = q_bf16;
            gp.t_bf16 = table_bf16;
            gp.dpad = pl.dpad;
            gp.n_qtiles = n_qtiles;
            gp.n_splits = pl.n_splits;
            gp.debug_skip_publish = -1;
            void* span = hmx_profile_begin("topk_seed_bf16", s);
            if (pl.KS == 4)
                hipLaunchKernelGGL(group_max_kernel<4>, dim3(n_qtiles * pl.n_splits), dim3(64), 0, s, gp, gm_tps,
                                   group_max_step(), gmax);
            else if (pl.KS == 8)
                hipLaunchKernelGGL(group_max_kernel<8>, dim3(n_qtiles * pl.n_splits), dim3(64), 0, s, gp, gm_tps,
                                   group_max_step(), gmax);
            else if (pl.KS == 2)
                hipLaunchKernelGGL(group_max_kernel<2>, dim3(n_qtiles * pl.n_splits), dim3(64), 0, s, gp, gm_tps,
                                   group_max_step(), gmax);
            else
                hipLaunchKernelGGL(group_max_kernel<1>, dim3(n_qtiles * pl.n_splits), dim3(64), 0, s, gp, gm_tps,
                                   group_max_step(), gmax);
            hmx_profile_end(span, s);
            HMX_CHECK_LAUNCH();
            hipLaunchKernelGGL(group_threshold_kernel, dim3((unsigned)rows_pad), dim3(64), 0, s, gmax, 4 * pl.n_splits,
                               queries + q0 * d, table, nq, n_rows, (int)d, (int)k, excl ? exclude + q0 * n_excl : nullptr,
                               excl ? (int)n_excl : 0, id_offset, max_row_norm, init_thr);
            HMX_CHECK_LAUNCH();
            thr0 = init_thr;
        }
        for (int lvl = 0; lvl < (group_seed ? 0 : tp.n_seed); ++lvl) {
            // exact top-K of the prefix lands in the output buffers (overwritten by the next pass)
            const int rc = pass(tp.seed[lvl], tp.seed_rows[lvl], thr0, nullptr, "topk_seed_bf16");
            if (rc != HMX_OK) return rc;
            hipLaunchKernelGGL(seed_threshold_kernel, dim3((unsigned)hmx_ceil_div(rows_pad, 256)), dim3(256), 0, s,
                               out_scores + q0 * k, queries + q0 * d, nq, rows_pad, (int)d, (int)k, max_row_norm,
                               init_thr);
            HMX_CHECK_LAUNCH();
            thr0 = init_thr;
        }
        if (samples && ext_thr != nullptr && sample_out == nullptr) {
            // start thresholds from the caller (e.g. the largest of the sampled thresholds of all item ranges)
            if (hipMemsetAsync(init_thr, 0, (size_t)rows_pad * sizeof(float), s) != hipSuccess ||
                hipMemcpyAsync(init_thr, ext_thr + q0, (size_t)nq * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
                return HMX_ELAUNCH;
            thr0 = init_thr;
        } else if (samples) {
            // many queries, every wave walks the whole table: sampled warm-up thresholds (see warm_threshold_kernel)
            const int64_t full_tiles = n_rows / 32;
            // 1/32 of the tiles, but not fewer than 4096 (short item ranges need relatively more: the start-up work
            // of a pass depends on the absolute number of rows seen), and never more than 1/6 (measured on a 500 K-row range: 1/4 21 ms, 1/6 17 ms, 1/8 18 ms, 1/16 20 ms, 1/32 21 ms)
            int n_sample = (int)std::min<int64_t>(std::max<int64_t>(full_tiles / 32, std::min<int64_t>(4096, full_tiles / 6)), 8192);
            if (const char* e = getenv("HMX_TOPK_WARM_TILES")) {     // diagnostic override of the sample size
                const int64_t v = atoll(e);
                if (v >= 64 && v <= full_tiles) n_sample = (int)v;
            }
            if (sample_out != nullptr && sample_ranges > 1) {
                // this range's sample is one of `sample_ranges` that the caller combines (largest threshold wins):
                // the combined sample is sized like that of one table of the combined length
                const int64_t all_tiles = full_tiles * sample_ranges;
                const int64_t all = std::min<int64_t>(std::max<int64_t>(all_tiles / 32, std::min<int64_t>(4096, all_tiles / 6)), 8192);
                n_sample = (int)std::min<int64_t>(std::max<int64_t>(hmx_ceil_div(all * sample_boost(), sample_ranges), 64), full_tiles);
            }
            if (n_sample >= 64) {
                const int stride_tiles = (int)((full_tiles - 1) / (n_sample - 1));
                CoarseParams wp{};
                wp.q_bf16 = q_bf16;
                wp.t_bf16 = table_bf16;
                wp.dpad = pl.dpad;
                wp.debug_skip_publish = -1;
                void* span = hmx_profile_begin("topk_warm_bf16", s);
                if (sample_out != nullptr && sample_values > 1)
                    launch_warm(pl.NB, pl.KS, k <= 10 ? 5 : 8, wp, n_qtiles, n_sample, stride_tiles,
                                sample_out + q0 * sample_values, s, sample_values / 2, nq);
                else
                    launch_warm(pl.NB, pl.KS, k <= 10 ? 5 : 8, wp, n_qtiles, n_sample, stride_tiles, init_thr, s);
                hmx_profile_end(span, s);
                HMX_CHECK_LAUNCH();
                thr0 = init_thr;
            }
            if (sample_out != nullptr) {
                if (thr0 == nullptr) {
                    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)hmx_ceil_div(nq * sample_values, 256)), dim3(256), 0, s,
                                       sample_out + q0 * sample_values, nq * sample_values, kNegInf);
                    HMX_CHECK_LAUNCH();
                } else if (sample_values == 1) {
                    if (hipMemcpyAsync(sample_out + q0, init_thr, (size_t)nq * sizeof(float), hipMemcpyDeviceToDevice, s) !=
                        hipSuccess)
                        return HMX_ELAUNCH;
                }
            }
        }
        if (sample_out != nullptr) continue;
        const int rc = pass(pl, n_rows, thr0, uncertain ? uncertain + q0 : nullptr,
                            bf16 ? "topk_coarse_bf16" : "topk_coarse_f32");
        if (rc != HMX_OK) return rc;
    }
    return HMX_OK;
}

}  // namespace

extern "C" int hmx_table_prepare_bf16(const float* table, int64_t n_rows, int64_t d, uint16_t* table_bf16,
                                      float* max_row_norm, void* stream) {
    HMX_ENTER();
    if (!table || !table_bf16 || !max_row_norm || n_rows <= 0 || d <= 0 || d > 128) return HMX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(max_row_norm, 0, sizeof(float), s) != hipSuccess) return HMX_ELAUNCH;
    hipLaunchKernelGGL(to_bf16_padded_kernel, dim3((unsigned)std::min<int64_t>(hmx_ceil_div(n_rows, 16), 8192)), dim3(256), 0, s, table,
                       n_rows, (int)d, n_rows, dpad_for(d), table_bf16, reinterpret_cast<unsigned*>(max_row_norm));
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" size_t hmx_score_topk_workspace_bytes(int64_t n_q, int64_t n_rows, int64_t d, int64_t k) {
    TopkPlans a, b;
    size_t need = 1536;
    if (make_plans(n_q, n_rows, d, k, true, a)) need = std::max(need, a.total());
    if (make_plans(n_q, n_rows, d, k, false, b)) need = std::max(need, b.total());
    return need;
}

extern "C" int hmx_score_topk_f32(const float* queries, int64_t n_q, const float* table, int64_t n_rows, int64_t d,
                                  int64_t k, const int64_t* exclude, int64_t n_excl, int64_t id_offset,
                                  int64_t* out_ids, float* out_scores, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    HMX_ENTER();
    return run_topk(false, queries, n_q, table, nullptr, nullptr, n_rows, d, k, exclude, n_excl, id_offset, out_ids,
                    out_scores, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int hmx_score_topk_f32_flagged(const float* queries, int64_t n_q, const float* table, int64_t n_rows, int64_t d,
                                          int64_t k, const int64_t* exclude, int64_t n_excl, int64_t id_offset,
                                          const int32_t* flags, int64_t* out_ids, float* out_scores, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (!flags && n_q > 0) return HMX_EINVAL;
    return run_topk(false, queries, n_q, table, nullptr, nullptr, n_rows, d, k, exclude, n_excl, id_offset, out_ids,
                    out_scores, nullptr, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, 1, 1,
                    flags);
}

extern "C" int hmx_score_topk_bf16(const float* queries, int64_t n_q, const float* table, const uint16_t* table_bf16,
                                   const float* max_row_norm, int64_t n_rows, int64_t d, int64_t k,
                                   const int64_t* exclude, int64_t n_excl, int64_t id_offset, int64_t* out_ids,
                                   float* out_scores, int32_t* uncertain, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    HMX_ENTER();
    return run_topk(true, queries, n_q, table, table_bf16, max_row_norm, n_rows, d, k, exclude, n_excl, id_offset,
                    out_ids, out_scores, uncertain, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int hmx_topk_sample_thresholds_bf16(const float* queries, int64_t n_q, const uint16_t* table_bf16,
                                               int64_t n_rows, int64_t d, int64_t k, int64_t n_ranges,
                                               int64_t n_values, float* thresholds, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if ((!thresholds && n_q > 0) || n_ranges < 1) return HMX_EINVAL;
    // one value, or the m <= 5 (k <= 10; 8 otherwise) largest sampled values of each half
    if (n_values != 1 && (n_values < 2 || n_values % 2 || n_values / 2 > (k <= 10 ? 5 : 8))) return HMX_EINVAL;
    return run_topk(true, queries, n_q, nullptr, table_bf16, nullptr, n_rows, d, k, nullptr, 0, 0, nullptr, nullptr,
                    nullptr, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr, thresholds, n_ranges,
                    (int)n_values);
}

extern "C" int hmx_kth_largest_f32(const float* vals, int64_t n_slabs, int64_t slab_stride, int64_t n_q, int64_t n_values,
                                   int64_t k, float* out, void* stream) {
    HMX_ENTER();
    if (n_q == 0) return HMX_OK;
    if (!vals || !out || n_slabs <= 0 || n_values <= 0 || k <= 0 || k > 16 || n_q < 0) return HMX_EINVAL;
    hipLaunchKernelGGL(kth_largest_kernel, dim3((unsigned)hmx_ceil_div(n_q, 256)), dim3(256), 0, (hipStream_t)stream, vals,
                       (int)n_slabs, slab_stride, n_q, (int)n_values, (int)k, out);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}

extern "C" int hmx_score_topk_bf16_seeded(const float* queries, int64_t n_q, const float* table,
                                          const uint16_t* table_bf16, const float* max_row_norm, int64_t n_rows,
                                          int64_t d, int64_t k, int64_t id_offset, const float* start_thresholds,
                                          int64_t* out_ids, float* out_scores, float* bounds, int32_t* failed,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    HMX_ENTER();
    if (n_q > 0 && (!bounds || !failed)) return HMX_EINVAL;
    return run_topk(true, queries, n_q, table, table_bf16, max_row_norm, n_rows, d, k, nullptr, 0, id_offset, out_ids,
                    out_scores, failed, workspace, workspace_bytes, (hipStream_t)stream, start_thresholds, bounds);
}

extern "C" int hmx_debug_ring_skip_publish(int64_t stage) {
    g_debug_skip_publish = stage >= 0 && stage < ((int64_t)1 << 30) ? (int)stage : -1;
    return HMX_OK;
}

extern "C" int hmx_topk_merge(const int64_t* cand_ids, const float* cand_scores, int64_t n_shards, int64_t n_q,
                              int64_t k, int64_t* out_ids, float* out_scores, void* stream) {
    HMX_ENTER();
    if (n_q == 0) return HMX_OK;
    if (!cand_ids || !cand_scores || !out_ids || !out_scores || n_shards <= 0 || k <= 0 || n_shards * k > 1024)
        return HMX_EINVAL;
    if (k <= 16)
        hipLaunchKernelGGL(topk_merge_insert_kernel, dim3((unsigned)hmx_ceil_div(n_q, 256)), dim3(256), 0,
                           (hipStream_t)stream, cand_ids, cand_scores, (int)n_shards, n_q, (int)k, out_ids, out_scores);
    else
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)n_q), dim3(64), 0, (hipStream_t)stream, cand_ids,
                           cand_scores, (int)n_shards, n_q, (int)k, out_ids, out_scores);
    HMX_CHECK_LAUNCH();
    return HMX_OK;
}
