// Ranking kernel: ascending order of all candidate costs, equal costs in
// candidate order.
//
// Replaces myqsort(lambda, arfitness, arindex, ...) (lmcma_path_planner/src/lmcma.cpp:91-104,
// called at :315).  The reference's comparator looks at the value only and libc's
// merge sort keeps equal values in input order; here every key is the 64-bit
// word (order-preserving image of the float cost << 32) | candidate index, so keys
// are distinct and any correct sort yields that same permutation.
//
// Populations up to 16384 are ranked by counting, which uses every CU: block
// (it, jt) counts, for its 256 candidates, the keys of j-tile jt that are below the
// candidate's own key (the j-tile is read from LDS at one address per step, a
// broadcast); k_rank_scatter adds the per-tile counts, which is the candidate's
// rank, and scatters candidate and cost to that position.  O(lambda^2) compares,
// but 16.7 M of them at lambda = 4096 is ~1 us of the chip, where a one-workgroup
// sorting network occupies a single CU for tens of us.
//
// Larger populations fall back to sorted tiles plus a merge:
// Step 1: each workgroup of 512 threads sorts one tile of 4096 keys with a bitonic
// network, 8 keys per thread.  Element i lives in thread i / 8, register i % 8, so
// of the 78 compare-exchange rounds 33 stay inside a thread, 39 cross lanes of one
// wave (shuffles, no barrier) and only the 6 rounds with stride >= 512 go through
// LDS with workgroup barriers.
// Step 2 (only when there are several tiles): every element adds, over the other
// tiles, the number of keys below its own (binary search) to its position in its
// tile: that sum is its global rank, and it scatters itself there.
#include <stdlib.h>

#include "device_types.hpp"

namespace lmcma {

namespace {

constexpr int kThreads = 512;
constexpr int kPerThread = kSortTile / kThreads;  // 8
static_assert(kPerThread == 8, "the in-thread rounds below are written for 8 keys per thread");

__device__ __forceinline__ uint32_t order_bits(float f)
{
    if (f != f) f = __builtin_inff();  // NaN ranks with +inf
    f += 0.0f;                          // -0 compares equal to +0 in the reference
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float order_bits_inv(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// after the call a <= b when up, a >= b otherwise
__device__ __forceinline__ void cx(uint64_t &a, uint64_t &b, bool up)
{
    const bool swap = (a > b) == up;
    const uint64_t lo = swap ? b : a, hi = swap ? a : b;
    a = lo;
    b = hi;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
    const uint32_t lo = __shfl_xor((uint32_t)v, mask, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(kThreads) void k_sort_tiles(const float *__restrict__ cost, int lambda,
                                                         uint64_t *__restrict__ keys_out,
                                                         float *__restrict__ sorted_cost,
                                                         int32_t *__restrict__ perm, int single_tile)
{
    __shared__ uint64_t xch[kSortTile];  // [register][thread]: conflict-free 8-byte columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int base = blockIdx.x * kSortTile;
    const int i0 = tid * kPerThread;

    uint64_t key[kPerThread];
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        const int c = base + i0 + e;
        key[e] = (c < lambda) ? (((uint64_t)order_bits(cost[c]) << 32) | (uint32_t)c)
                              : ~0ull;  // padding sorts last
    }

    for (int size = 2; size <= kSortTile; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= kThreads) {
                // partner thread is in another wave: exchange through LDS
                const int other = tid ^ (stride / kPerThread);
                const bool keep_min = ((i0 & size) == 0) == ((tid & (stride / kPerThread)) == 0);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < kPerThread; ++e) xch[e * kThreads + tid] = key[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < kPerThread; ++e) {
                    const uint64_t o = xch[e * kThreads + other];
                    key[e] = keep_min ? (o < key[e] ? o : key[e]) : (o > key[e] ? o : key[e]);
                }
            } else if (stride >= kPerThread) {
                // partner thread is another lane of this wave
                const int mask = stride / kPerThread;
                const bool keep_min = ((i0 & size) == 0) == ((lane & mask) == 0);
#pragma unroll
                for (int e = 0; e < kPerThread; ++e) {
                    const uint64_t o = shfl_xor_u64(key[e], mask);
                    key[e] = keep_min ? (o < key[e] ? o : key[e]) : (o > key[e] ? o : key[e]);
                }
            } else {
                // both elements are registers of this thread; direction from bit `size` of i
                const bool up_thread = (i0 & size) == 0;  // valid when size >= 8
                if (stride == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cx(key[e], key[e + 4], size >= 8 ? up_thread : true);
                } else if (stride == 2) {
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {
                        const bool up = size >= 8 ? up_thread : ((h & size) == 0);
                        cx(key[h], key[h + 2], up);
                        cx(key[h + 1], key[h + 3], up);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const bool up = size >= 8 ? up_thread : ((e & size) == 0);
                        cx(key[e], key[e + 1], up);
                    }
                }
            }
        }
    }

#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        const int i = i0 + e;
        if (single_tile) {
            if (i < lambda) {
                sorted_cost[i] = order_bits_inv((uint32_t)(key[e] >> 32));
                perm[i] = (int32_t)(uint32_t)key[e];
            }
        } else {
            keys_out[base + i] = key[e];
        }
    }
}

__global__ __launch_bounds__(256) void k_merge_rank(const uint64_t *__restrict__ keys, int ntiles,
                                                    float *__restrict__ sorted_cost,
                                                    int32_t *__restrict__ perm)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ntiles * kSortTile) return;
    const uint64_t key = keys[e];
    if (key == ~0ull) return;  // padding
    const int mine = e / kSortTile;
    int rank = e - mine * kSortTile;
    for (int b = 0; b < ntiles; ++b) {
        if (b == mine) continue;
        const uint64_t *tile = keys + (size_t)b * kSortTile;
        int lo = 0, hi = kSortTile;  // first position whose key is >= mine
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tile[mid] < key) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    sorted_cost[rank] = order_bits_inv((uint32_t)(key >> 32));
    perm[rank] = (int32_t)(uint32_t)key;
}

// ---- rank by counting ----------------------------------------------------------------

template <int JT>
__global__ __launch_bounds__(256) void k_rank_count(const float *__restrict__ cost, int lambda,
                                                    uint32_t *__restrict__ counts /* [njt][lambda] */)
{
    __shared__ uint64_t jkeys[JT];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * JT;
    for (int t = threadIdx.x; t < JT; t += 256) {
        const int j = j0 + t;
        jkeys[t] = (j < lambda) ? (((uint64_t)order_bits(cost[j]) << 32) | (uint32_t)j) : ~0ull;
    }
    const uint64_t mine = (i < lambda) ? (((uint64_t)order_bits(cost[i]) << 32) | (uint32_t)i) : 0ull;
    __syncthreads();
    uint32_t below = 0;
#pragma unroll 16
    for (int t = 0; t < JT; ++t) below += (jkeys[t] < mine) ? 1u : 0u;
    if (i < lambda) counts[(size_t)blockIdx.y * lambda + i] = below;
}

// block-wide sum of one 64-bit count per thread, added to *accum by one atomic
__device__ __forceinline__ void block_add_u64(unsigned long long v, unsigned long long *accum,
                                              unsigned long long *s_wave /* [4] */)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (total) atomicAdd(accum, total);
    }
}

// number of entries of the ascending list prev[0..n) that are below c
__device__ __forceinline__ int count_below(const float *prev, int n, float c)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (prev[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Second half of ranking by counting, 64 candidates per block:
//   rank = sum of the per-tile counts; the candidate and its cost go to that position;
//   the population-success-rule count against last generation's list (staged in LDS);
//   and, when weights are given, the block's share of the elite recombination
//   (lmcma.cpp:321-326): sum over its own candidates with rank < mu of
//   weights[rank] * row, candidates in index order, as one row of block_partials.
// The reference adds the elites in rank order; this adds the same terms grouped by
// candidate index, which changes fp64 rounding only.
constexpr int kFinishCands = 64;

__global__ __launch_bounds__(256) void k_rank_finish(const float *__restrict__ cost, int lambda,
                                                     const uint32_t *__restrict__ counts, int njt,
                                                     float *__restrict__ sorted_cost,
                                                     int32_t *__restrict__ perm,
                                                     const OptScalars *__restrict__ opt,
                                                     const float *__restrict__ prev_sorted,
                                                     unsigned long long *__restrict__ psr_accum,
                                                     const float *__restrict__ rows, int ld, int N,
                                                     int first_candidate, int n_local,
                                                     const double *__restrict__ weights, int mu,
                                                     double *__restrict__ block_partials, int first_block)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    // carve: recombination slots [nslots][ld] doubles | ranks [64] ints | previous list [lambda] floats
    const int nq = ld >> 2;
    const int nslots = weights ? 256 / nq : 0;
    double *s_part = s_dyn;
    int *s_rank = reinterpret_cast<int *>(s_dyn + (size_t)nslots * ld);
    float *s_prev = reinterpret_cast<float *>(s_rank + kFinishCands);
    __shared__ unsigned long long s_wave[4];

    const bool psr = opt != nullptr && opt->itr > 0;  // uniform over the grid
    if (psr) {
        const float4 *src = reinterpret_cast<const float4 *>(prev_sorted);
        float4 *dst = reinterpret_cast<float4 *>(s_prev);
        for (int t = threadIdx.x; t < ((lambda + 3) >> 2); t += 256) dst[t] = src[t];
    }
    const int c0 = blockIdx.x * kFinishCands;
    // four threads per candidate add the per-tile counts
    {
        const int t = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int i = c0 + t;
        uint32_t rank = 0;
        if (i < lambda)
            for (int jt = part; jt < njt; jt += 4) rank += counts[(size_t)jt * lambda + i];
        rank += __shfl_xor(rank, 1, 64);
        rank += __shfl_xor(rank, 2, 64);
        if (part == 0) s_rank[t] = (i < lambda) ? (int)rank : 0x7fffffff;
    }
    __syncthreads();
    unsigned long long below = 0ull;
    if (threadIdx.x < kFinishCands) {
        const int i = c0 + threadIdx.x;
        if (i < lambda) {
            const int rank = s_rank[threadIdx.x];
            const float c = order_bits_inv(order_bits(cost[i]));  // NaN -> +inf, -0 -> +0 like the keys
            sorted_cost[rank] = c;
            perm[rank] = i;
            if (psr) below = (unsigned long long)count_below(s_prev, lambda, c);
        }
    }
    if (weights) {
        const int slot = threadIdx.x / nq, q = threadIdx.x % nq;
        if (slot < nslots) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int t = slot; t < kFinishCands; t += nslots) {
                const int rank = s_rank[t];
                const int local = c0 + t - first_candidate;
                if (rank < mu && local >= 0 && local < n_local) {
                    const float4 x = reinterpret_cast<const float4 *>(rows + (size_t)local * ld)[q];
                    const double w = weights[rank];
                    a0 += w * (double)x.x; a1 += w * (double)x.y;
                    a2 += w * (double)x.z; a3 += w * (double)x.w;
                }
            }
            double *dst = s_part + (size_t)slot * ld + 4 * q;
            dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
        }
    }
    if (psr) block_add_u64(below, psr_accum, s_wave);  // contains a barrier
    else __syncthreads();
    if (weights) {
        const int lb = (int)blockIdx.x - first_block;  // blocks without local candidates write nothing
        const bool has_local = c0 + kFinishCands > first_candidate && c0 < first_candidate + n_local;
        if (has_local)
            for (int n = threadIdx.x; n < N; n += 256) {
                double acc = 0.0;
                for (int sl = 0; sl < nslots; ++sl) acc += s_part[(size_t)sl * ld + n];
                block_partials[(size_t)lb * N + n] = acc;
            }
    }
}

// the same count for the sorting paths: one thread per candidate, list in global memory
__global__ __launch_bounds__(256) void k_psr_count(const float *__restrict__ cost, int lambda,
                                                   const OptScalars *__restrict__ opt,
                                                   const float *__restrict__ prev_sorted,
                                                   unsigned long long *__restrict__ psr_accum)
{
    __shared__ unsigned long long s_wave[4];
    if (opt->itr <= 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long below =
        (i < lambda) ? (unsigned long long)count_below(prev_sorted, lambda, order_bits_inv(order_bits(cost[i]))) : 0ull;
    block_add_u64(below, psr_accum, s_wave);
}

}  // namespace

size_t select_workspace_bytes(int lambda)
{
    const size_t tiles = (size_t)select_num_tiles(lambda) * kSortTile * sizeof(uint64_t);
    if (lambda > kRankCountMax) return tiles;
    const int jt = lambda <= 4096 ? 128 : 512;
    const size_t counts = (size_t)((lambda + jt - 1) / jt) * lambda * sizeof(uint32_t);
    return counts > tiles ? counts : tiles;
}

int select_num_tiles(int lambda) { return (lambda + kSortTile - 1) / kSortTile; }

bool select_can_recombine(int lambda)
{
    return lambda <= kRankCountMax && getenv("LMCMA_HIP_SELECT_SORT") == nullptr;
}

int select_recombine_rows(int first_candidate, int n_local)
{
    return (first_candidate + n_local + kFinishCands - 1) / kFinishCands - first_candidate / kFinishCands;
}

hipError_t launch_select(hipStream_t s, const float *cost_all, int lambda, uint64_t *keys_ws,
                         float *sorted_cost, int32_t *perm, const OptScalars *opt,
                         const float *prev_sorted, unsigned long long *psr_accum,
                         const RecombineArgs *rc)
{
    if (lambda <= 0) return hipErrorInvalidValue;
    if (!prev_sorted || !psr_accum) opt = nullptr;
    if (lambda <= kRankCountMax && getenv("LMCMA_HIP_SELECT_SORT") == nullptr) {
        uint32_t *counts = reinterpret_cast<uint32_t *>(keys_ws);
        const int it = (lambda + 255) / 256;
        int njt;
        if (lambda <= 4096) {
            njt = (lambda + 127) / 128;
            hipLaunchKernelGGL(k_rank_count<128>, dim3(it, njt), dim3(256), 0, s, cost_all, lambda, counts);
        } else {
            njt = (lambda + 511) / 512;
            hipLaunchKernelGGL(k_rank_count<512>, dim3(it, njt), dim3(256), 0, s, cost_all, lambda, counts);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int nblocks = (lambda + kFinishCands - 1) / kFinishCands;
        const int nslots = rc ? 256 / (rc->ld >> 2) : 0;
        const size_t lds = (rc ? (size_t)nslots * rc->ld * sizeof(double) : 0) + kFinishCands * sizeof(int) +
                           (opt ? (size_t)((lambda + 3) & ~3) * sizeof(float) : 0);
        static size_t configured = 48 * 1024;
        if (lds > configured) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rank_finish),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured = lds;
        }
        hipLaunchKernelGGL(k_rank_finish, dim3(nblocks), dim3(256), lds, s, cost_all, lambda, counts, njt,
                           sorted_cost, perm, opt, prev_sorted, psr_accum, rc ? rc->rows : nullptr,
                           rc ? rc->ld : 4, rc ? rc->N : 0, rc ? rc->first_candidate : 0, rc ? rc->n_local : 0,
                           rc ? rc->weights : nullptr, rc ? rc->mu : 0, rc ? rc->block_partials : nullptr,
                           rc ? rc->first_candidate / kFinishCands : 0);
        return hipGetLastError();
    }
    if (rc) return hipErrorInvalidValue;  // the fused recombination exists only on the counting path
    const int ntiles = select_num_tiles(lambda);
    hipLaunchKernelGGL(k_sort_tiles, dim3(ntiles), dim3(kThreads), 0, s, cost_all, lambda, keys_ws,
                       sorted_cost, perm, ntiles == 1 ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ntiles > 1) {
        const int total = ntiles * kSortTile;
        hipLaunchKernelGGL(k_merge_rank, dim3((total + 255) / 256), dim3(256), 0, s, keys_ws, ntiles,
                           sorted_cost, perm);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (opt) {
        hipLaunchKernelGGL(k_psr_count, dim3((lambda + 255) / 256), dim3(256), 0, s, cost_all, lambda, opt,
                           prev_sorted, psr_accum);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace lmcma
