// Ranking by counting for populations up to 16384, and the dispatcher of the ranking
// paths.  Ascending order of all candidate costs, equal costs in candidate order.
//
// Replaces myqsort(lambda, arfitness, arindex, ...) (lmcma_path_planner/src/lmcma.cpp:91-104,
// called at :315), the head of LMCMA::update (:316-326, weighted recombination) and the
// rank sums of the population success rule (:393-413).
//
// One launch, every CU busy:
//   * every workgroup stages all keys in LDS; 16 lanes share one candidate and count
//     the keys below it (ties by candidate index), 4 keys per lane and LDS read; the sum
//     over the 16 lanes is the candidate's rank, and candidate and cost are scattered to
//     that position.  O(lambda^2) compares, but 16.7 M of them at lambda = 4096 is about
//     a microsecond of the chip, where a one-workgroup sorting network occupies a single
//     CU for tens of microseconds;
//   * the same 16 lanes find the number of last generation's costs below the candidate
//     with a 16-way search of last generation's ascending list (also staged in LDS); the
//     workgroup's total goes to psr_parts[workgroup];
//   * the workgroup adds weights[rank] * row over its own elite candidates, candidates
//     in index order, and writes one row of block_partials.  The reference adds the
//     elites in rank order; grouping by candidate index changes fp64 rounding only.
#include <stdlib.h>

#include <algorithm>

#include "update_body.hpp"

namespace lmcma {

namespace {

struct RankArgs {
    const float *cost;
    int lambda;
    float *sorted_cost;
    int32_t *perm;
    const OptScalars *opt;       // with prev_sorted: success-rule count when opt->itr > 0
    const float *prev_sorted;
    uint32_t *psr_parts;         // [gridDim.x]
    const float *rows;           // recombination (weights == nullptr: none)
    int ld, N, first_candidate, n_local;
    const double *weights;
    int mu;
    int weights_in_lds;
    double *block_partials;      // [workgroups holding local candidates][N]
    int first_block;
    // PRE_EMAX > 0: the last workgroup of the grid runs the first part of the update instead
    UpdateArgs pre;
    int pre_threads;
    Batch batch;
    int tile_len;                // > 0: every tile of this many keys is ranked among itself
};

__device__ __forceinline__ uint32_t row16_sum_u32(uint32_t v)
{
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0xB1>(v);
    return v;
}

// LDS layout (doubles first): recombination slots, candidate weights, weight table, then
// keys, previous list, per-wave counts
size_t rank_lds_bytes(int TB, int lambda, int N, int mu, bool psr, bool recombine, bool weights_in_lds)
{
    const size_t lpad = ((size_t)lambda + 255) & ~(size_t)255;
    size_t b = 0;
    if (recombine)
        b += ((size_t)(TB > N ? TB : N) + TB / 16 + (weights_in_lds ? (((size_t)mu + 1) & ~(size_t)1) : 0)) * sizeof(double);
    b += lpad * sizeof(uint32_t);
    if (psr) b += lpad * sizeof(float);
    b += 16 * sizeof(uint32_t);
    return b;
}

#ifdef LMCMA_DIAG
#define RANK_STAMP(i) st[i] = __builtin_amdgcn_s_memrealtime()
#else
#define RANK_STAMP(i)
#endif

template <int TB, int PRE_EMAX>
__global__ __launch_bounds__(TB) void k_rank(const RankArgs r_in)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    if (PRE_EMAX > 0) {
        // The vectors of the direction store that do not depend on this generation's costs are
        // rebuilt here, on a CU the ranking leaves idle, instead of after the ranking.
        if (blockIdx.x == gridDim.x - 1) {
            update_pre_body<(PRE_EMAX > 0 ? PRE_EMAX : 8)>(shifted(r_in.pre, blockIdx.y), s_dyn, r_in.pre_threads);
            return;
        }
    }
    RankArgs r = r_in;
    if (r.batch.stride != 0) {
        const size_t off = (size_t)blockIdx.y * r.batch.stride;  // query blockIdx.y of a batch
        shift_ptr(r.cost, off); shift_ptr(r.sorted_cost, off); shift_ptr(r.perm, off); shift_ptr(r.opt, off);
        shift_ptr(r.prev_sorted, off); shift_ptr(r.psr_parts, off); shift_ptr(r.rows, off); shift_ptr(r.weights, off);
        shift_ptr(r.block_partials, off);
    }
    int bx = blockIdx.x;
    if (r.tile_len > 0) {
        // tiles: this workgroup's tile is a population of its own, its list and its (tile-local)
        // permutation go to the tile's stretch of the outputs
        const int bpt = r.tile_len / (TB / 16), tile = bx / bpt;
        bx -= tile * bpt;
        r.cost += (size_t)tile * r.tile_len;
        r.sorted_cost += (size_t)tile * r.tile_len;
        r.perm += (size_t)tile * r.tile_len;
        r.lambda = r.tile_len;
    }
    constexpr int CPB = TB / 16;  // candidates per workgroup
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    const int lambda = r.lambda, N = r.N;
    const int lpad = (lambda + 255) & ~255;  // keys padded to whole rounds of 256
    const bool recombine = r.weights != nullptr;
    const bool psr = r.prev_sorted != nullptr && r.opt != nullptr && r.opt->itr > 0;  // uniform over the grid
    const bool wlds = recombine && r.weights_in_lds;

    double *s_part = s_dyn;                                                   // [rg][N]
    double *s_w = s_part + (recombine ? (TB > N ? TB : N) : 0);               // [CPB] weight of each candidate
    double *s_weights = s_w + (recombine ? CPB : 0);                          // [mu] when staged
    uint32_t *s_key = reinterpret_cast<uint32_t *>(s_weights + (wlds ? ((r.mu + 1) & ~1) : 0));
    float *s_prev = reinterpret_cast<float *>(s_key + lpad);                  // [lpad] when psr
    uint32_t *s_below = reinterpret_cast<uint32_t *>(s_prev + (psr ? lpad : 0));  // [TB / 64]
#ifdef LMCMA_DIAG
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
#endif
    RANK_STAMP(0);

    // ---- stage keys, last generation's list and the weights.  The costs were written by
    // the previous kernel, so the first touch of every line misses this XCD's L2: all loads
    // of the first batch (4 x 16 bytes per lane and array, and the first rows of the
    // recombination) are issued before anything waits for one of them.
    const int n4 = lpad >> 2, live4 = (lambda + 3) >> 2, n2 = (r.mu + 1) >> 1;
    const float4 *ksrc = reinterpret_cast<const float4 *>(r.cost);  // padded to a multiple of 16 bytes
    const float4 *psrc = reinterpret_cast<const float4 *>(r.prev_sorted);
    const double2 *wsrc = reinterpret_cast<const double2 *>(r.weights);
    uint4 *kdst = reinterpret_cast<uint4 *>(s_key);
    float4 *pdst = reinterpret_cast<float4 *>(s_prev);
    double2 *wdst = reinterpret_cast<double2 *>(s_weights);
    const int c0 = bx * CPB;
    const RowSumPlan pl = row_sum_plan(N, TB);
    const int ch = (CPB + pl.rg - 1) / pl.rg;  // candidates per row group
    const int rgi = tid / pl.cw, n0 = tid % pl.cw;
    {
        float4 kc[4], pv[4];
        double2 wv[4];
        float xr0[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + u * TB;
            kc[u] = ksrc[t < live4 ? t : live4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + u * TB;
            pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (psr) pv[u] = psrc[t < live4 ? t : live4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + u * TB;
            wv[u] = make_double2(0.0, 0.0);
            if (wlds) wv[u] = wsrc[t < n2 ? t : n2 - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xr0[u] = 0.f;
            if (recombine) {
                int local = c0 + (rgi < pl.rg ? rgi : 0) * ch + u - r.first_candidate;
                local = local < 0 ? 0 : (local >= r.n_local ? r.n_local - 1 : local);
                xr0[u] = r.rows[(size_t)local * r.ld + (n0 < N ? n0 : N - 1)];
            }
        }
        asm volatile("" ::: "memory");  // the loads above stay above: none is sunk into a branch below
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + u * TB, j = 4 * t;
            if (t < n4) {
                uint4 k;  // padding (~0) ranks after every cost
                k.x = j < lambda ? order_bits(kc[u].x) : ~0u;
                k.y = j + 1 < lambda ? order_bits(kc[u].y) : ~0u;
                k.z = j + 2 < lambda ? order_bits(kc[u].z) : ~0u;
                k.w = j + 3 < lambda ? order_bits(kc[u].w) : ~0u;
                kdst[t] = k;
            }
            if (psr && t < live4) pdst[t] = pv[u];
            if (wlds && t < n2) wdst[t] = wv[u];
        }
        // populations beyond 16 lanes x 4 x TB: the rest, batch by batch
        for (int t0 = tid + 4 * TB; t0 < n4; t0 += TB) {
            const int j = 4 * t0;
            const float4 c = ksrc[t0 < live4 ? t0 : live4 - 1];
            uint4 k;
            k.x = j < lambda ? order_bits(c.x) : ~0u;
            k.y = j + 1 < lambda ? order_bits(c.y) : ~0u;
            k.z = j + 2 < lambda ? order_bits(c.z) : ~0u;
            k.w = j + 3 < lambda ? order_bits(c.w) : ~0u;
            kdst[t0] = k;
            if (psr && t0 < live4) pdst[t0] = psrc[t0];
        }
        if (wlds)
            for (int t0 = tid + 4 * TB; t0 < n2; t0 += TB) wdst[t0] = wsrc[t0];
        __syncthreads();
        RANK_STAMP(1);

        // ---- count the keys below each candidate's own.  A wave owns four candidates; every
        // lane reads four keys of its own per round (no LDS word is read twice by a wave) and
        // compares them with the wave's four candidates.
        const int w4 = (tid >> 6) << 2;  // first of this wave's candidates, within the workgroup
        uint32_t mine_q[4], cq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) mine_q[q] = __builtin_amdgcn_readfirstlane(s_key[c0 + w4 + q]);  // < lpad
        {
            const int r0 = c0 >> 8, rounds = lpad >> 8;  // CPB divides 256: one round holds the whole workgroup
            const uint4 *k4 = reinterpret_cast<const uint4 *>(s_key) + wl;  // round t: keys 256 t + 4 wl ..
            // rounds before the candidates' own: every key there has a lower index, so ties count
#pragma unroll 2
            for (int t = 0; t < r0; ++t) {
                const uint4 a = k4[64 * t];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cq[q] += (a.x <= mine_q[q]) + (a.y <= mine_q[q]) + (a.z <= mine_q[q]) + (a.w <= mine_q[q]);
            }
            {
                const uint4 a = k4[64 * r0];
                const int j = 256 * r0 + 4 * wl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t mq = mine_q[q];
                    const int iq = c0 + w4 + q;
                    cq[q] += (a.x < mq || (a.x == mq && j < iq));
                    cq[q] += (a.y < mq || (a.y == mq && j + 1 < iq));
                    cq[q] += (a.z < mq || (a.z == mq && j + 2 < iq));
                    cq[q] += (a.w < mq || (a.w == mq && j + 3 < iq));
                }
            }
#pragma unroll 2
            for (int t = r0 + 1; t < rounds; ++t) {
                const uint4 a = k4[64 * t];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cq[q] += (a.x < mine_q[q]) + (a.y < mine_q[q]) + (a.z < mine_q[q]) + (a.w < mine_q[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cq[q] = add_lane_xor32(add_lane_xor16(row16_sum_u32(cq[q])));
        }
        // the 16 lanes of group g carry on with candidate c0 + g
        const int gq = grp & 3;
        const int rank = (int)(gq == 0 ? cq[0] : gq == 1 ? cq[1] : gq == 2 ? cq[2] : cq[3]);
        const uint32_t mine = gq == 0 ? mine_q[0] : gq == 1 ? mine_q[1] : gq == 2 ? mine_q[2] : mine_q[3];
        const int i = c0 + grp;
        const bool live = i < lambda;
        const float value = order_bits_inv(mine);  // NaN -> +inf, -0 -> +0 like the keys
        RANK_STAMP(2);

        // ---- success rule: number of last generation's costs below this one, 16-way search
        uint32_t below = 0;
        if (psr) {
            int lo = 0, len = lambda;  // the answer lies in [lo, lo + len]
            while (len > 0) {
                const int chunk = (len + 15) >> 4;
                const int p = lo + (lane + 1) * chunk - 1;
                const bool lt = p < lo + len && s_prev[p] < value;
                const unsigned long long mask = __ballot(lt);
                const int q = __popcll((mask >> (tid & 48)) & 0xFFFFull);  // probes below, of this group's 16
                const int end = lo + len;
                lo += q * chunk;
                len = chunk - 1 < end - lo ? chunk - 1 : end - lo;
            }
            below = (live && lane == 0) ? (uint32_t)lo : 0u;
        }
        // the four groups of a wave add their counts: lanes 0, 16, 32, 48 hold them
        below = add_lane_xor32(add_lane_xor16(below));
        if (lane == 0) {
            if (wl == 0) s_below[tid >> 6] = below;
            if (recombine) {
                const int local = i - r.first_candidate;
                const bool elite = live && rank < r.mu && local >= 0 && local < r.n_local;
                s_w[grp] = elite ? (wlds ? s_weights[rank] : r.weights[rank]) : 0.0;
            }
        }
        __syncthreads();
        RANK_STAMP(3);
        uint32_t below_total = 0;
        if (tid == 0) {
#pragma unroll
            for (int w = 0; w < TB / 64; ++w) below_total += s_below[w];
        }

        // ---- elite rows of this workgroup, candidates in index order
        if (recombine) {
            const bool has_local = c0 + CPB > r.first_candidate && c0 < r.first_candidate + r.n_local;
            if (rgi < pl.rg) {
                for (int n = n0; n < N; n += pl.cw) {
                    double acc = 0.0;
                    for (int u0 = 0; u0 < ch; u0 += 8) {
                        float x[8];
                        if (u0 == 0 && n == n0) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) x[u] = xr0[u];
                        } else {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                int local = c0 + rgi * ch + u0 + u - r.first_candidate;
                                local = local < 0 ? 0 : (local >= r.n_local ? r.n_local - 1 : local);
                                x[u] = r.rows[(size_t)local * r.ld + n];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int c = rgi * ch + u0 + u;
                            const double w = (u0 + u < ch && c < CPB) ? s_w[c] : 0.0;
                            if (w != 0.0) acc += w * (double)x[u];
                        }
                    }
                    s_part[rgi * N + n] = acc;
                }
            }
            __syncthreads();
            if (has_local) {
                double *dst = r.block_partials + (size_t)((int)blockIdx.x - r.first_block) * N;
                for (int n = tid; n < N; n += TB) dst[n] = row_sums_total(s_part, N, n, pl.rg);
            }
        }
        // the scattered results last: a workgroup barrier waits for the stores issued before it
        if (lane == 0 && live) {
            r.sorted_cost[rank] = value;
            r.perm[rank] = i;
        }
        if (tid == 0) r.psr_parts[blockIdx.x] = below_total;
    }
#ifdef LMCMA_DIAG
    RANK_STAMP(4);
    if (tid == 0 && blockIdx.x == gridDim.x - 1 - (PRE_EMAX > 0 ? 1 : 0))
        for (int q = 0; q < 5; ++q) g_update_stamps[8 + q] = st[q];
#endif
}

// ---- one workgroup ranks one whole population: a sample sort ------------------------------
// Counting costs lambda^2 compares per population: fine when one population has the chip to
// itself (k_rank: a microsecond or two on every CU), wasteful when many populations are ranked
// by one launch (several queries in one context), where all that matters is the CU time a
// population takes.  Here 512 threads rank up to 4096 keys with some lambda (9 + lambda / 256)
// compares: 512 keys sampled at a fixed stride are ranked among themselves, the 511 lowest
// split the key space into 512 buckets; every key finds its bucket by binary search and takes a
// place in it (LDS counter); inside a bucket, 16 lanes count for every member the members below
// it.  rank = start of the bucket + that count.  Keys are the 64-bit words (order-preserving
// image of the cost << 32) | candidate index, all distinct, so equal costs rank in candidate
// order and equal costs cannot crowd one bucket.  The success-rule count is a binary search of
// every cost in last generation's list (staged in LDS).  The elite rows are not added here.
constexpr int kSortedThreads = 512;
constexpr int kSortedMax = 4096;      // 8 keys per thread
constexpr int kSortedBuckets = 512;

typedef double SortKey;
__device__ __forceinline__ SortKey sort_key(uint32_t bits, uint32_t index) { return (double)bits * 4096.0 + (double)index; }

size_t rank_sorted_lds_bytes(bool psr)
{
    return (size_t)kSortedMax * (sizeof(SortKey) + sizeof(int) + sizeof(uint16_t)) + (2 * kSortedBuckets + 8) * sizeof(SortKey) +
           2 * kSortedBuckets * sizeof(int) + 16 * sizeof(uint32_t) + (psr ? kSortedMax * sizeof(float) : 0);
}

template <int PRE_EMAX>
__global__ __launch_bounds__(kSortedThreads) void k_rank_sorted(const RankArgs r_in)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    if (PRE_EMAX > 0) {
        if (blockIdx.x == 1) {  // the first part of this query's update, beside its ranking
            update_pre_body<(PRE_EMAX > 0 ? PRE_EMAX : 8)>(shifted(r_in.pre, blockIdx.y), s_dyn, r_in.pre_threads);
            return;
        }
    }
    RankArgs r = r_in;
    if (r.batch.stride != 0) {
        const size_t off = (size_t)blockIdx.y * r.batch.stride;
        shift_ptr(r.cost, off); shift_ptr(r.sorted_cost, off); shift_ptr(r.perm, off); shift_ptr(r.opt, off);
        shift_ptr(r.prev_sorted, off); shift_ptr(r.psr_parts, off);
    }
    constexpr int TB = kSortedThreads, KPT = kSortedMax / kSortedThreads, NB = kSortedBuckets;
    static_assert(NB == TB, "one sample, one bucket per thread");
    const int tid = threadIdx.x;
    const int lambda = r.lambda;
    const bool psr = r.prev_sorted != nullptr && r.opt != nullptr && r.opt->itr > 0;

    // A key is the integer (order-preserving image of the cost) * 4096 + candidate index, below
    // 2^44, held as a double: exact, and compared by the fp64 compare instead of a 64-bit integer one.
    SortKey *s_grp = reinterpret_cast<SortKey *>(s_dyn);        // [4096 + 8] keys grouped by bucket, then +inf
    SortKey *s_samp = s_grp + kSortedMax + 8;                   // [NB] the sampled keys
    SortKey *s_split = s_samp + NB;                             // [NB] ... in ascending order
    int *s_rank = reinterpret_cast<int *>(s_split + NB);        // [4096] rank of the key at a grouped position
    int *s_hist = s_rank + kSortedMax;                          // [NB] members per bucket
    int *s_start = s_hist + NB;                                 // [NB] first grouped position of a bucket
    uint16_t *s_bkt = reinterpret_cast<uint16_t *>(s_start + NB);  // [4096] bucket of a grouped position
    uint32_t *s_below = reinterpret_cast<uint32_t *>(s_bkt + kSortedMax);  // [TB / 64]
    float *s_prev = reinterpret_cast<float *>(s_below + 16);    // [4096] when psr

#ifdef LMCMA_DIAG
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RANK_STAMP(0);
    // ---- loads: this thread's keys (candidates tid, tid + 512, ...) and last generation's list
    float c[KPT];
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        const int i = tid + e * TB;
        c[e] = r.cost[i < lambda ? i : lambda - 1];
    }
    if (psr) {
        const int n4 = lambda >> 2;  // lambda is a multiple of 512
        float4 pv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * TB;
            pv[u] = reinterpret_cast<const float4 *>(r.prev_sorted)[t < n4 ? t : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * TB;
            if (t < n4) reinterpret_cast<float4 *>(s_prev)[t] = pv[u];
        }
    }
    s_hist[tid] = 0;
    if (tid < 8) s_grp[lambda + tid] = 1.7976931348623157e308;
    SortKey key[KPT];
    const int stride = lambda / NB;  // NB samples
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        const int i = tid + e * TB;
        key[e] = sort_key(order_bits(c[e]), (uint32_t)i);
    }
    // sample j is candidate j * stride + stride / 2
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        const int i = tid + e * TB;
        // (stride is a power of two or not: i - stride / 2 divisible by stride)
        const int q = (i - (stride >> 1)) / stride;
        if (i < lambda && q * stride + (stride >> 1) == i) s_samp[q] = key[e];
    }
    __syncthreads();
    RANK_STAMP(1);

    // ---- the samples in ascending order.  Every wave sorts its 64 (a bitonic network over the
    // lanes, 21 exchanges), then every sample adds its positions in the other waves' sorted runs
    // (binary searches) to its position in its own: its rank among all of them.
    {
        SortKey mine = s_samp[tid];
        const int ln = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int strd = size >> 1; strd > 0; strd >>= 1) {
                const SortKey o = __shfl_xor(mine, strd, 64);
                const bool keep_min = ((ln & size) == 0) == ((ln & strd) == 0);
                mine = keep_min ? fmin(mine, o) : fmax(mine, o);
            }
        }
        __syncthreads();  // every thread has read its sample
        s_samp[tid] = mine;  // run wv, position ln
        __syncthreads();
        constexpr int NW = TB / 64;
        int cnt[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) cnt[w] = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            SortKey o[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) o[w] = s_samp[64 * w + cnt[w] + step - 1];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int w = 0; w < NW; ++w) cnt[w] += (o[w] < mine) ? step : 0;
        }
        // (five steps reach 63 of a run's 64; the last entry of each run is compared on its own)
        int below = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const SortKey last = s_samp[64 * w + 63];
            const int c = cnt[w] + ((cnt[w] == 63 && last < mine) ? 1 : 0);
            below += (w == wv) ? ln : c;
        }
        s_split[below] = mine;
    }
    __syncthreads();
    RANK_STAMP(2);

    // ---- bucket of every key = number of the NB - 1 lowest samples below it; a place in the bucket
    int place[KPT];
    {
        int cnt[KPT];
#pragma unroll
        for (int e = 0; e < KPT; ++e) cnt[e] = 0;
#pragma unroll
        for (int step = NB >> 1; step > 0; step >>= 1) {
            SortKey sp[KPT];
#pragma unroll
            for (int e = 0; e < KPT; ++e) sp[e] = s_split[cnt[e] + step - 1];
            asm volatile("" ::: "memory");  // the eight reads are in flight together
#pragma unroll
            for (int e = 0; e < KPT; ++e) cnt[e] += (sp[e] < key[e]) ? step : 0;
        }
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            place[e] = 0;
            if (tid + e * TB < lambda) place[e] = (cnt[e] << 16) | atomicAdd(&s_hist[cnt[e]], 1);
        }
    }
    __syncthreads();
    RANK_STAMP(3);
    if (tid < 64) {
        // exclusive prefix over the bucket sizes, NB / 64 consecutive ones per lane
        constexpr int PL = NB / 64;
        int cn[PL], sum = 0;
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            cn[u] = s_hist[PL * tid + u];
            sum += cn[u];
        }
        int incl = sum;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(incl, sft, 64);
            if (tid >= sft) incl += o;
        }
        int at = incl - sum;
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            s_start[PL * tid + u] = at;
            at += cn[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        if (tid + e * TB < lambda) {
            const int b = place[e] >> 16;
            place[e] = s_start[b] + (place[e] & 0xFFFF);
            s_grp[place[e]] = key[e];
            s_bkt[place[e]] = (uint16_t)b;
        }
    }
    __syncthreads();
    RANK_STAMP(4);

    // ---- ranks inside the buckets.  A thread takes grouped positions tid, tid + 512, ...: the
    // lanes of a wave hold neighbouring positions, so they walk through a handful of buckets
    // and one LDS word serves many lanes.
#pragma unroll 1
    for (int e = 0; e < KPT; ++e) {
        const int p = tid + e * TB;
        if (p < lambda) {
            const int b = s_bkt[p];
            const int base = s_start[b], n = s_hist[b];
            const SortKey mine = s_grp[p];
            // whole batches of eight reads: what follows a bucket in the grouped array are the later
            // buckets' keys (and eight entries of +inf after the last), none of them below this key
            int below = 0;
            for (int k = 0; k < n; k += 8) {
                SortKey o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = s_grp[base + k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) below += o[u] < mine;
            }
            s_rank[p] = base + below;
        }
    }
    __syncthreads();
    RANK_STAMP(5);

    // ---- the sorted list in LDS (over the grouped keys, which are done with): every key to its rank
    float *s_val = reinterpret_cast<float *>(s_grp);            // [4096] costs in ascending order
    int *s_idx = reinterpret_cast<int *>(s_val + kSortedMax);   // [4096] their candidates
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
        const int i = tid + e * TB;
        if (i < lambda) {
            const int rank = s_rank[place[e]];
            s_val[rank] = order_bits_inv(order_bits(c[e]));
            s_idx[rank] = i;
        }
    }
    __syncthreads();
    RANK_STAMP(6);
    // out in whole lines: thread t writes ranks 8 t .. 8 t + 7
    if (KPT * tid < lambda) {
        const float4 v0 = reinterpret_cast<const float4 *>(s_val)[2 * tid], v1 = reinterpret_cast<const float4 *>(s_val)[2 * tid + 1];
        const int4 i0 = reinterpret_cast<const int4 *>(s_idx)[2 * tid], i1 = reinterpret_cast<const int4 *>(s_idx)[2 * tid + 1];
        reinterpret_cast<float4 *>(r.sorted_cost)[2 * tid] = v0;
        reinterpret_cast<float4 *>(r.sorted_cost)[2 * tid + 1] = v1;
        reinterpret_cast<int4 *>(r.perm)[2 * tid] = i0;
        reinterpret_cast<int4 *>(r.perm)[2 * tid + 1] = i1;
    }

    // ---- success rule: the number of pairs (cost of last generation, cost of this one) with the
    // former below the latter, by merging the two ascending lists: thread t merges elements
    // [t per, (t + 1) per) of the merged order (a current cost goes before an equal previous one)
    // and adds, for every current cost it takes, the previous costs already taken.
    uint32_t below_prev = 0;
    if (psr) {
        const int per = (2 * lambda) / TB, d = tid * per;
        int lo = d > lambda ? d - lambda : 0, hi = d < lambda ? d : lambda;
        while (__any(lo < hi)) {
            const bool act = lo < hi;
            const int mid = act ? (lo + hi) >> 1 : 0;
            const int jb = act ? d - mid - 1 : 0;
            const float a = s_val[mid], b = s_prev[jb];
            if (act) {
                if (a <= b) lo = mid + 1;
                else hi = mid;
            }
        }
        int ia = lo, jb = d - lo;
        float a = s_val[ia < lambda ? ia : lambda - 1], b = s_prev[jb < lambda ? jb : lambda - 1];
        for (int t = 0; t < per; ++t) {
            const bool take_a = jb >= lambda || (ia < lambda && a <= b);
            below_prev += take_a ? (uint32_t)jb : 0u;
            ia += take_a ? 1 : 0;
            jb += take_a ? 0 : 1;
            const float na = s_val[ia < lambda ? ia : lambda - 1], nb = s_prev[jb < lambda ? jb : lambda - 1];
            a = na;
            b = nb;
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) below_prev += __shfl_xor(below_prev, sft, 64);
    if ((tid & 63) == 0) s_below[tid >> 6] = below_prev;
    __syncthreads();
    if (tid == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < TB / 64; ++w) total += s_below[w];
        r.psr_parts[0] = total;
    }
#ifdef LMCMA_DIAG
    RANK_STAMP(7);
    if (tid == 0 && blockIdx.y == 0)
        for (int q = 0; q < 8; ++q) g_update_stamps[8 + q] = st[q];
#endif
}

// ---- one shard of a population: global ranks of the local candidates by merging ----------
// In a job of G contexts each context first ranks its own n_local candidates among
// themselves (k_rank on the local costs) and puts its ASCENDING costs into its block of the
// exchanged cost vector.  After the exchange every context holds G sorted lists, and the
// global rank of its element at local position p is
//     p + sum over lower contexts of #{their costs <= x} + sum over higher contexts of #{their costs < x}
// (ties across contexts resolve by context order = candidate index order, like the stable
// sort of the whole population).  The success rule needs the number of pairs (previous cost
// j, current cost i) with j < i, over the whole population; a context counts the pairs of ITS
// previous costs (last generation's own list, kept as keys): for its previous element y,
// lambda - sum over all contexts of #{their current costs <= y}.  Per element that is 2 G
// searches, each a binary search among the list's pivots (every 16th key, in LDS) and one
// 64-byte window of the list read by four of the element's 16 lanes, instead of lambda compares:
// the work per context no longer grows with the square of the population, and a workgroup reads
// lambda / 16 keys plus the windows of its 32 elements, not the lists.  sorted_cost / perm receive the entries of the
// local candidates only; the workgroup's elite rows go to one row of block_partials.
struct RankMergeArgs {
    const float *lists;           // [G][n_local] ascending costs of every context (the exchanged vector)
    int lambda, n_local, me;      // lambda = G n_local; me = index of this context
    const int32_t *local_perm;    // [n_local] local candidate at local position p
    float *sorted_cost;
    int32_t *perm;
    const OptScalars *opt;        // success-rule count when opt->itr > 0
    const uint32_t *prev_own;     // [n_local] this context's list of last generation, as keys
    uint32_t *own_out;            // [n_local] this context's list of this generation, as keys
    uint32_t *psr_parts;          // [gridDim.x]
    float *best_out;              // lowest cost of the population
    const float *rows;
    int ld, N;
    const double *weights;
    int mu;
    double *block_partials;       // [gridDim.x][N]
    int owners;                   // 1, or G: this launch ranks the elements of every list (see MergeArgs)
};

constexpr int kMergeChunk = 32768;  // the longest list; also the largest population one context ranks in tiles
constexpr int kMergeMaxLambda = 1 << 19;  // 32768 pivots = 128 KiB of LDS
constexpr int kMergeWindowLog = 4;  // a pivot every 16 keys: one window = 4 lanes x 16 bytes

// number of entries of the ascending key list s[0..n) that are below x (or_equal: at or below):
// a binary search by one lane.  The 16 lanes of a group each take one of the searches their
// element needs (a list each), so a search costs a dozen dependent LDS reads but no
// cross-lane traffic.
__device__ __forceinline__ int count_below(const uint32_t *s, int n, uint32_t x, bool or_equal)
{
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        const uint32_t k = s[lo + half];
        const bool go = or_equal ? k <= x : k < x;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}

// The windows of one element's searches j0 .. j0 + cnt - 1, four searches per batch (lane u of its
// 16-lane group holds the window of search j0 + u): lanes 4 q .. 4 q + 3 read the 16 keys of the
// window of the batch's search q, 16 bytes each; all NB loads are requested before the first is
// used and none behind a branch (a search that does not exist or has nothing to count reads
// window 0 of its list and adds nothing).  A window instruction of a wave thus serves 16 searches.
struct MergeWindowArgs {
    const float *lists;
    int nl, G, me, j0, cnt, lane, w;
    uint32_t mine, yprev;
};

template <int NB>
__device__ __forceinline__ void merge_windows(const MergeWindowArgs &a, int &rank, int &at_or_below_prev)
{
    constexpr int LPW = 1 << (kMergeWindowLog - 2), SPB = 16 / LPW;  // lanes per window, searches per batch
    const int sq = a.lane / LPW, sub = a.lane % LPW;
    float4 win[NB];
    int wu[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int u = SPB * b + sq;
        wu[b] = __shfl(a.w, u, 16);
        const int ju = a.j0 + (u < a.cnt ? u : 0), g = ju >= a.G ? ju - a.G : ju;
        const int wc = wu[b] < 0 ? 0 : wu[b];
        win[b] = reinterpret_cast<const float4 *>(a.lists + (size_t)g * a.nl + ((size_t)wc << kMergeWindowLog))[sub];
    }
    asm volatile("" ::: "memory");
    // (nothing computed from a window is moved up between the requests)
#pragma unroll
    for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(win[b].x), "+v"(win[b].y), "+v"(win[b].z), "+v"(win[b].w));
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int u = SPB * b + sq, ju = a.j0 + u;
        const bool prev = ju >= a.G;
        const int g = prev ? ju - a.G : ju;
        const uint32_t x = prev ? a.yprev : a.mine;
        const bool or_equal = prev || g < a.me;
        // "at or below x" = "below x + 1", unless x is the largest key, which every key is at or below
        const uint32_t t = or_equal ? x + 1u : x;
        const bool all = or_equal && x == ~0u;
        int n = (order_bits(win[b].x) < t) + (order_bits(win[b].y) < t) + (order_bits(win[b].z) < t) + (order_bits(win[b].w) < t);
        n = all ? 4 : n;
        if (sub == 0) n += wu[b] << kMergeWindowLog;
        n = (u < a.cnt && wu[b] >= 0) ? n : 0;
        at_or_below_prev += prev ? n : 0;
        rank += prev ? 0 : n;
    }
}

// doubles of LDS for the partial sums of the elite rows: one padded row per row group
__host__ __device__ inline size_t merge_part_doubles(int TB, int ld)
{
    const int c4 = ld >> 2, cw4 = c4 < TB ? c4 : TB;
    return (size_t)(TB / cw4) * ld;
}

template <int TB>
__global__ __launch_bounds__(TB) void k_rank_merge(const RankMergeArgs r)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    constexpr int CPB = TB / 16;
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    const int lambda = r.lambda, N = r.N, nl = r.n_local;
    const int G = lambda / nl;
    const bool psr = r.opt != nullptr && r.opt->itr > 0;
    // the list this workgroup's elements belong to, and its workgroup index within that list
    const int bpl = (nl + CPB - 1) / CPB;
    const int me = r.owners > 1 ? (int)blockIdx.x / bpl : r.me;
    const int bx = r.owners > 1 ? (int)blockIdx.x % bpl : (int)blockIdx.x;
    const size_t own0 = r.owners > 1 ? (size_t)me * nl : 0;  // where list `me` starts in the per-owner arrays

    double *s_part = s_dyn;                               // [row groups][ld]
    int *s_rank = reinterpret_cast<int *>(s_part + merge_part_doubles(TB, r.ld));  // [CPB] global rank of each element
    int *s_cand = s_rank + CPB;                           // [CPB] local candidate of each element
    uint32_t *s_below = reinterpret_cast<uint32_t *>(s_cand + CPB);  // [TB / 64], then the elite mask
    uint32_t *s_elite = s_below + 15;                     // bit c: element c of the workgroup is an elite
    uint32_t *s_key = s_below + 16;                       // [G][n_local / 16] pivots
    if (tid == 0) *s_elite = 0u;                          // (a barrier follows before the first use)

#ifdef LMCMA_DIAG
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
#endif
    RANK_STAMP(0);
    // A workgroup takes CPB / 2 positions from the low end of its list and the mirrored CPB / 2
    // from the high end: the elites are the low positions of every list, and this way every
    // workgroup adds up about the same number of elite rows.  n_local is a multiple of CPB.
    static_assert(CPB <= 32 && CPB % 2 == 0, "one mask word per workgroup");
    const int p = grp < CPB / 2 ? bx * (CPB / 2) + grp : nl - (bx + 1) * (CPB / 2) + (grp - CPB / 2);
    const bool live = p < nl;
    const int pc = live ? p : nl - 1;
    const uint32_t mine = order_bits(r.lists[(size_t)me * nl + pc]);
    const uint32_t yprev = psr ? r.prev_own[own0 + pc] : 0u;
    const int cand = r.local_perm[own0 + pc];

    // ---- every 16th key of every list (a list's pivots) to LDS: lambda / 16 scattered 4-byte reads
    const int np = nl >> kMergeWindowLog;                 // n_local is a multiple of the window
    for (int t0 = tid; t0 < G * np; t0 += 4 * TB) {
        float pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * TB < G * np ? t0 + u * TB : tid;
            const int g = t / np, j = t - g * np;
            pv[u] = r.lists[(size_t)g * nl + ((size_t)j << kMergeWindowLog)];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + u * TB < G * np) s_key[t0 + u * TB] = order_bits(pv[u]);
    }
    __syncthreads();
    RANK_STAMP(1);
    uint32_t kmin = ~0u;
    if (blockIdx.x == 0 && tid == 0)
        for (int g = 0; g < G; ++g) kmin = min(kmin, s_key[(size_t)g * np]);  // a list's first entry is its lowest
    // Searches of this element: its own cost in every other list (ties across contexts resolve by
    // context order: lists before this context's count their equal keys), and, for the success
    // rule, its previous cost in every list.  Lane j takes search j among the pivots: with c pivots
    // below x (at or below), every key before window c - 1 is below x and none after it is, so the
    // count is 16 (c - 1) + the keys of that one window below x.  The group then reads the windows
    // of up to 16 searches, four at a time.
    int rank = lane == 0 ? pc : 0, at_or_below_prev = 0;  // summed over the group's lanes afterwards
    const int nsearch = psr ? 2 * G : G;
    for (int j0 = 0; j0 < nsearch; j0 += 16) {
        const int j = j0 + lane;
        int w = -1;                                       // window of lane's search; -1: nothing to count
        if (j < nsearch) {
            const bool prev = j >= G;
            const int g = prev ? j - G : j;
            if (prev || g != me)
                w = count_below(s_key + (size_t)g * np, np, prev ? yprev : mine, prev || g < me) - 1;
        }
        const int cnt = nsearch - j0;
        RANK_STAMP(5);
        const MergeWindowArgs wa{r.lists, nl, G, me, j0, cnt, lane, w, mine, yprev};
        constexpr int SPB = 64 >> kMergeWindowLog;        // searches per batch of window reads
        if (cnt <= SPB) merge_windows<1>(wa, rank, at_or_below_prev);
        else if (cnt <= 2 * SPB) merge_windows<2>(wa, rank, at_or_below_prev);
        else if (cnt <= 4 * SPB) merge_windows<4>(wa, rank, at_or_below_prev);
        else merge_windows<16 / SPB>(wa, rank, at_or_below_prev);
    }
    // every lane of the group gets the sums over its 16 lanes
    rank = (int)row16_sum_u32((uint32_t)rank);
    at_or_below_prev = (int)row16_sum_u32((uint32_t)at_or_below_prev);
    RANK_STAMP(2);
    const float value = order_bits_inv(mine);
    // pairs (this context's previous element, any current cost above it)
    uint32_t bsum = (psr && live && lane == 0) ? (uint32_t)(lambda - at_or_below_prev) : 0u;
    bsum = add_lane_xor32(add_lane_xor16(bsum));
    if (wl == 0) s_below[tid >> 6] = bsum;
    if (lane == 0) {
        const bool elite = live && rank < r.mu;
        s_rank[grp] = rank;
        s_cand[grp] = cand;
        if (elite) atomicOr(s_elite, 1u << grp);
    }
    __syncthreads();
    RANK_STAMP(3);
    uint32_t below_total = 0;
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < TB / 64; ++w) below_total += s_below[w];
    }
    // ---- elite rows of this workgroup, elements in local order.  A thread owns four adjacent
    // coordinates (one 16-byte load per row) of the elites e = rgi, rgi + rg, ... of the workgroup;
    // all its loads are requested before the first is used.
    const int ld = r.ld, c4 = ld >> 2;                    // ld is a multiple of 4; columns N..ld-1 are padding
    const int cw4 = c4 < TB ? c4 : TB, rg = TB / cw4;
    const int rgi = tid / cw4, col0 = tid - rgi * cw4;
    const uint32_t elite_mask = *s_elite;
    const int ne = __popc(elite_mask);
    if (rgi < rg) {
        for (int col = col0; col < c4; col += cw4) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            uint32_t left = elite_mask;
            for (int k = 0; k < rgi && left != 0u; ++k) left &= left - 1u;  // skip to elite number rgi
            for (int e0 = rgi; e0 < ne; e0 += 8 * rg) {
                float4 x[8];
                double w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    w[u] = 0.0;
                    x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (left != 0u) {
                        const int c = __ffs((int)left) - 1;
                        w[u] = r.weights[s_rank[c]];      // (requested together with the row)
                        x[u] = reinterpret_cast<const float4 *>(r.rows + (own0 + (size_t)s_cand[c]) * ld)[col];
                        for (int k = 0; k < rg && left != 0u; ++k) left &= left - 1u;  // the next one of this thread
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (w[u] != 0.0) {
                        a0 += w[u] * (double)x[u].x;
                        a1 += w[u] * (double)x[u].y;
                        a2 += w[u] * (double)x[u].z;
                        a3 += w[u] * (double)x[u].w;
                    }
            }
            double *dst = s_part + (size_t)rgi * ld + 4 * col;
            reinterpret_cast<double2 *>(dst)[0] = make_double2(a0, a1);
            reinterpret_cast<double2 *>(dst)[1] = make_double2(a2, a3);
        }
    }
    __syncthreads();
    {
        double *dst = r.block_partials + (size_t)blockIdx.x * N;
        for (int n = tid; n < N; n += TB) dst[n] = row_sums_total(s_part, ld, n, rg);
    }
    if (lane == 0 && live) {
        r.sorted_cost[rank] = value;
        r.perm[rank] = me * nl + cand;
        r.own_out[own0 + p] = mine;  // the owner's list, for the next generation's success rule
    }
    if (tid == 0) {
        r.psr_parts[blockIdx.x] = below_total;
        if (blockIdx.x == 0) *r.best_out = order_bits_inv(kmin);
    }
#ifdef LMCMA_DIAG
    RANK_STAMP(4);
    if (tid == 0 && blockIdx.x == gridDim.x - 1)
        for (int q = 0; q < 6; ++q) g_update_stamps[8 + q] = st[q];
#endif
}

template <int TB>
hipError_t launch_rank_merge_t(hipStream_t s, const RankMergeArgs &r, int blocks)
{
    constexpr int CPB = TB / 16;
    const int keys = r.lambda >> kMergeWindowLog;  // the pivots of all lists
    const size_t lds = merge_part_doubles(TB, r.ld) * sizeof(double) + (size_t)2 * CPB * sizeof(int) +
                       (size_t)(16 + keys) * sizeof(uint32_t);
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_rank_merge<TB>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_rank_merge<TB>), dim3(blocks), dim3(TB), lds, s, r);
    return hipGetLastError();
}

template <int TB, int PRE_EMAX>
hipError_t launch_rank_tp(hipStream_t s, const RankArgs &r, int blocks, size_t lds)
{
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_rank<TB, PRE_EMAX>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_rank<TB, PRE_EMAX>), dim3(blocks + (PRE_EMAX > 0 ? 1 : 0), r.batch.queries), dim3(TB), lds, s, r);
    return hipGetLastError();
}

// pre_emax: 0, or the register tile (8 / 10 / 12 / 16 coordinates per lane) of the update part riding along
template <int TB>
hipError_t launch_rank_t(hipStream_t s, const RankArgs &r, int blocks, size_t lds, int pre_emax)
{
    // the update part must fit the registers the workgroup size leaves per lane
    if (pre_emax == 8) return launch_rank_tp<TB, 8>(s, r, blocks, lds);
    if (pre_emax == 10) return launch_rank_tp<TB, 10>(s, r, blocks, lds);
    if (pre_emax == 12) return launch_rank_tp<TB, 12>(s, r, blocks, lds);
    if (pre_emax == 16) return launch_rank_tp<TB, 16>(s, r, blocks, lds);
    if (pre_emax != 0) return hipErrorInvalidValue;  // a tile this kernel has no variant for
    return launch_rank_tp<TB, 0>(s, r, blocks, lds);
}

}  // namespace

#ifdef LMCMA_DIAG
extern "C" hipError_t lmcma_diag_rank_stamps(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_update_stamps), 16 * sizeof(unsigned long long));
}
#endif

bool select_by_counting(int lambda)
{
    static const bool forced_sort = getenv("LMCMA_HIP_SELECT_SORT") != nullptr;  // (read once, not per launch)
    return lambda <= kRankCountMax && !forced_sort;
}

// candidates per workgroup of the counting path: 16 lanes each
int select_candidates_per_block(int lambda)
{
    // 512 threads = 32 candidates per workgroup: measured best at lambda = 4096 (a smaller
    // workgroup counts faster but leaves twice as many partial rows for the one-workgroup
    // update to read, a larger one the other way round)
    int threads = lambda >= 1024 ? 512 : 256;
    static const int forced = getenv("LMCMA_HIP_RANK_THREADS") ? atoi(getenv("LMCMA_HIP_RANK_THREADS")) : 0;
    if (forced == 256 || forced == 512 || forced == 1024) threads = forced;
    return threads / 16;
}

int select_max_partial_rows(int first_candidate, int n_local)
{
    const int cpb = 16;  // the smallest workgroup of the counting path ranks 16 candidates
    return (first_candidate + n_local + cpb - 1) / cpb - first_candidate / cpb;
}

int rank_tile_len(int lambda)
{
    // above one tile of the counting rank and up to what the merge holds in LDS at once; equal tiles
    // that are whole rounds of the counting kernel
    // (up to two tiles the one-launch counting rank is faster: 11.8 us against 6.6 + 9.2 us at 8192)
    static const int tile = getenv("LMCMA_HIP_RANK_TILE") ? atoi(getenv("LMCMA_HIP_RANK_TILE")) : kSortTile;
    if (tile < 256 || tile > kSortTile) return 0;
    if (lambda <= 2 * tile || lambda > kMergeChunk || getenv("LMCMA_HIP_NO_TILED_RANK") != nullptr) return 0;
    const int tiles = (lambda + tile - 1) / tile;
    if (lambda % tiles != 0 || (lambda / tiles) % 256 != 0) return 0;
    return lambda / tiles;
}

bool rank_merge_possible(int lambda, int n_local, int first_candidate)
{
    static const bool no_merge = getenv("LMCMA_HIP_NO_MERGE_RANK") != nullptr;
    // (the pivots of all lists, lambda / 16 keys, have to fit in LDS beside the row sums)
    return n_local < lambda && n_local % 64 == 0 && lambda % n_local == 0 && first_candidate % n_local == 0 &&
           n_local <= kMergeChunk && lambda <= kMergeMaxLambda && !no_merge;
}

hipError_t launch_rank_merge(hipStream_t s, const MergeArgs &a, int *rows_out)
{
    // (a context's own tiles: owners = lambda / n_local, checked below; a shard: the merge's limits)
    if (a.owners <= 1 && !rank_merge_possible(a.lambda, a.n_local, a.first_candidate)) return hipErrorInvalidValue;
    if (a.n_local < 64 || a.n_local % 64 != 0 || a.lambda % a.n_local != 0 || a.n_local > kMergeChunk || a.lambda > kMergeMaxLambda)
        return hipErrorInvalidValue;
    RankMergeArgs r{};
    r.lists = a.lists;
    r.lambda = a.lambda; r.n_local = a.n_local; r.me = a.first_candidate / a.n_local;
    r.local_perm = a.local_perm;
    r.sorted_cost = a.sorted_cost;
    r.perm = a.perm;
    r.opt = a.opt;
    r.prev_own = a.prev_own;
    r.own_out = a.own_out;
    r.psr_parts = a.psr_parts;
    r.best_out = a.best_out;
    r.rows = a.rows; r.ld = a.ld; r.N = a.N;
    r.weights = a.weights; r.mu = a.mu;
    r.block_partials = a.block_partials;
    r.owners = a.owners > 1 ? a.owners : 1;
    if (r.owners > 1 && (r.owners != a.lambda / a.n_local || a.first_candidate != 0)) return hipErrorInvalidValue;
    const int blocks = r.owners * ((a.n_local + 31) / 32);
    if (rows_out) *rows_out = blocks;
    return launch_rank_merge_t<512>(s, r, blocks);
}

hipError_t launch_select(hipStream_t s, const SelectArgs &a, SelectResult *res)
{
    SelectResult out{};
    if (res) *res = out;
    const int lambda = a.lambda;
    if (lambda <= 0) return hipErrorInvalidValue;

    if (a.tile_len > 0) {
        if (a.tile_len > kRankCountMax || lambda % a.tile_len != 0 || a.rc != nullptr || a.batch.queries > 1)
            return hipErrorInvalidValue;
    } else if (!select_by_counting(lambda)) {
        if (a.batch.queries > 1) return hipErrorInvalidValue;  // batches are ranked by counting only
        return launch_select_sort(s, a);
    }

    const OptScalars *opt = (a.prev_sorted && a.psr_accum) ? a.opt : nullptr;
    RankArgs r{};
    r.batch = a.batch.queries >= 1 ? a.batch : Batch{1, 0};
    // Many populations in one launch: one workgroup each (k_rank_sorted) instead of lambda / 32.
    // From 12 queries on the generation is shorter that way (lambda = 4096: 70.7 against 76.0 us with
    // 12 queries, 61.0 against 60.0 with 8; one query: 8.6 us by counting on every CU against some
    // 25 us for the single workgroup and the recombination launch it needs).
    static const int sorted_from = getenv("LMCMA_HIP_SORTED_RANK_FROM") ? atoi(getenv("LMCMA_HIP_SORTED_RANK_FROM")) : 12;
    if (a.tile_len == 0 && r.batch.queries >= sorted_from && lambda >= 1024 && lambda <= kSortedMax && lambda % kSortedBuckets == 0) {
        r.cost = a.cost_all;
        r.lambda = lambda;
        r.sorted_cost = a.sorted_cost;
        r.perm = a.perm;
        r.opt = opt;
        r.prev_sorted = opt ? a.prev_sorted : nullptr;
        r.psr_parts = reinterpret_cast<uint32_t *>(a.keys_ws);
        size_t lds = rank_sorted_lds_bytes(opt != nullptr);
        int pre_emax = 0;
        if (a.pre && update_is_split(a.pre->N, a.pre->m) && update_pre_rides_with_ranking(a.pre->m) &&
            update_pre_threads(a.pre->m) <= kSortedThreads && a.pre->N <= 256) {
            r.pre = *a.pre;
            r.pre_threads = update_pre_threads(a.pre->m);
            pre_emax = update_emax(a.pre->N);
            lds = std::max(lds, update_pre_lds_doubles(a.pre->N, a.pre->m) * sizeof(double));
        }
        const dim3 grid(pre_emax > 0 ? 2 : 1, r.batch.queries);
#define LMCMA_SORTED(E)                                                                                     \
    do {                                                                                                    \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_rank_sorted<E>), lds);          \
        if (e != hipSuccess) return e;                                                                      \
        hipLaunchKernelGGL((k_rank_sorted<E>), grid, dim3(kSortedThreads), lds, s, r);                      \
    } while (0)
        if (pre_emax == 8) LMCMA_SORTED(8);
        else if (pre_emax == 10) LMCMA_SORTED(10);
        else if (pre_emax == 12) LMCMA_SORTED(12);
        else if (pre_emax == 16) LMCMA_SORTED(16);
        else if (pre_emax == 0) LMCMA_SORTED(0);
        else return hipErrorInvalidValue;
#undef LMCMA_SORTED
        out.partial_rows = 0;  // the elite rows are added by launch_recombine
        out.pre_done = pre_emax > 0;
        out.psr_parts = opt ? 1 : 0;
        if (res) *res = out;
        return hipGetLastError();
    }
    r.tile_len = a.tile_len;
    r.cost = a.cost_all;
    r.lambda = lambda;
    r.sorted_cost = a.sorted_cost;
    r.perm = a.perm;
    r.opt = opt;
    r.prev_sorted = opt ? a.prev_sorted : nullptr;
    r.psr_parts = reinterpret_cast<uint32_t *>(a.keys_ws);
    const RecombineArgs *rc = a.rc;
    r.ld = 4;
    r.N = 4;
    const int keys = a.tile_len > 0 ? a.tile_len : lambda;  // keys a workgroup stages and counts over
    const int cpb = select_candidates_per_block(keys);
    const int TB = cpb * 16;
    const int blocks = (lambda + cpb - 1) / cpb;
    int rows = 0;
    if (rc) {
        r.rows = rc->rows; r.ld = rc->ld; r.N = rc->N;
        r.first_candidate = rc->first_candidate; r.n_local = rc->n_local;
        r.weights = rc->weights; r.mu = rc->mu;
        r.block_partials = rc->block_partials;
        r.first_block = rc->first_candidate / cpb;
        rows = (rc->first_candidate + rc->n_local + cpb - 1) / cpb - r.first_block;
    }
    const bool psr = opt != nullptr;
    size_t lds = rank_lds_bytes(TB, keys, r.N, r.mu, psr, rc != nullptr, true);
    r.weights_in_lds = rc && lds <= 150 * 1024;
    if (!r.weights_in_lds) lds = rank_lds_bytes(TB, keys, r.N, r.mu, psr, rc != nullptr, false);
    // The first part of the update as one more workgroup: when the ranking leaves CUs idle
    // (at most one workgroup per CU) and its workgroup is large enough for the vectors.
    int pre_emax = 0;
    if (a.pre && update_is_split(a.pre->N, a.pre->m) && update_pre_rides_with_ranking(a.pre->m) && blocks <= 256 && update_pre_threads(a.pre->m) <= TB &&
        a.pre->N <= 256) {  // longer vectors: the stand-alone kernel (fewer variants of this one)
        r.pre = *a.pre;
        r.pre_threads = update_pre_threads(a.pre->m);
        pre_emax = update_emax(a.pre->N);
        lds = std::max(lds, update_pre_lds_doubles(a.pre->N, a.pre->m) * sizeof(double));
    }
    hipError_t e;
    if (TB == 256) e = launch_rank_t<256>(s, r, blocks, lds, pre_emax);
    else if (TB == 512) e = launch_rank_t<512>(s, r, blocks, lds, pre_emax);
    else e = launch_rank_t<1024>(s, r, blocks, lds, pre_emax);
    out.partial_rows = rows;
    out.pre_done = pre_emax > 0;
    out.psr_parts = opt ? blocks : 0;
    if (res) *res = out;
    return e;
}

}  // namespace lmcma
