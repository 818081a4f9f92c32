// Ranking kernel: ascending order of all candidate costs, equal costs in
// candidate order.
//
// Replaces myqsort(lambda, arfitness, arindex, ...) (lmcma_path_planner/src/lmcma.cpp:91-104,
// called at :315).  The reference's comparator looks at the value only and libc's
// merge sort keeps equal values in input order; here every key is the 64-bit
// word (order-preserving image of the float cost << 32) | candidate index, so keys
// are distinct and any correct sort yields that same permutation.
//
// Populations up to 16384 are ranked by counting (k_rank.hip), which uses every CU.
// Larger populations are ranked here, by sorted tiles plus a merge:
// Step 1: each workgroup of 512 threads sorts one tile of 4096 keys with a bitonic
// network, 8 keys per thread.  Element i lives in thread i / 8, register i % 8, so
// of the 78 compare-exchange rounds 33 stay inside a thread, 39 cross lanes of one
// wave (shuffles, no barrier) and only the 6 rounds with stride >= 512 go through
// LDS with workgroup barriers.
// Step 2 (only when there are several tiles): every element adds, over the other
// tiles, the number of keys below its own (binary search) to its position in its
// tile: that sum is its global rank, and it scatters itself there.
#include <stdlib.h>

#include <mutex>

#include "device_funcs.hpp"

namespace lmcma {

namespace {

constexpr int kThreads = 512;
constexpr int kPerThread = kSortTile / kThreads;  // 8
static_assert(kPerThread == 8, "the in-thread rounds below are written for 8 keys per thread");

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
    // tiles of the sorting path; the counting path keeps its per-workgroup counts in the same buffer
    return (size_t)select_num_tiles(lambda) * kSortTile * sizeof(uint64_t);
}

int select_num_tiles(int lambda) { return (lambda + kSortTile - 1) / kSortTile; }

hipError_t launch_select_sort(hipStream_t s, const SelectArgs &a)
{
    const int lambda = a.lambda;
    const OptScalars *opt = (a.prev_sorted && a.psr_accum) ? a.opt : nullptr;
    const int ntiles = select_num_tiles(lambda);
    hipLaunchKernelGGL(k_sort_tiles, dim3(ntiles), dim3(kThreads), 0, s, a.cost_all, lambda, a.keys_ws,
                       a.sorted_cost, a.perm, ntiles == 1 ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ntiles > 1) {
        const int total = ntiles * kSortTile;
        hipLaunchKernelGGL(k_merge_rank, dim3((total + 255) / 256), dim3(256), 0, s, a.keys_ws, ntiles,
                           a.sorted_cost, a.perm);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (opt) {
        hipLaunchKernelGGL(k_psr_count, dim3((lambda + 255) / 256), dim3(256), 0, s, a.cost_all, lambda, opt,
                           a.prev_sorted, a.psr_accum);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace lmcma

// ---- ranking for callers that cannot allocate -------------------------------------------
// The sorted key tiles of the sorting path, for populations up to
// LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA, in a fixed array of the code object instead of a
// context's workspace: 1 MiB per GPU the library runs on, held until the process ends.
static_assert(LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA % lmcma::kSortTile == 0, "whole tiles");
static __device__ __align__(256) uint64_t g_rank_scratch[LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA];

namespace lmcma {

// Device address of g_rank_scratch on the calling thread's current device (every GPU holds
// its own copy of the array at its own address), looked up once per device.  Host code uses
// this pointer, never the array's name, wherever it needs the array's memory.
hipError_t rank_scratch_current_device(uint64_t **out)
{
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static uint64_t *ptrs[kMaxDevices] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_rank_scratch));
        if (e != hipSuccess) return e;
        ptrs[dev] = static_cast<uint64_t *>(p);
    }
    *out = ptrs[dev];
    return hipSuccess;
}

int set_last_error(int code, const char *msg);  // capi.hip

}  // namespace lmcma

extern "C" int lmcma_hip_rank_noalloc(const float *cost_dev, int lambda, float *sorted_cost_dev, int32_t *perm_dev,
                                      void *stream)
{
    using namespace lmcma;
    if (!cost_dev || !sorted_cost_dev || !perm_dev) return set_last_error(LMCMA_HIP_ERR_ARG, "rank_noalloc: null buffer");
    if (lambda < 1) return set_last_error(LMCMA_HIP_ERR_ARG, "rank_noalloc: lambda must be positive");
    if (lambda > LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA)
        return set_last_error(LMCMA_HIP_ERR_UNSUPPORTED,
                              "rank_noalloc: at most 131072 costs (LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA); larger "
                              "populations go through a context (lmcma_hip_select)");
    uint64_t *scratch = nullptr;
    if (rank_scratch_current_device(&scratch) != hipSuccess)
        return set_last_error(LMCMA_HIP_ERR_RUNTIME, "rank_noalloc: hipGetSymbolAddress failed");
    // every key of the tiles in use is written (padding included) before the merge reads it:
    // nothing of the last call's contents is read, nothing has to start at zero
    SelectArgs a{};
    a.cost_all = cost_dev;
    a.lambda = lambda;
    a.keys_ws = scratch;
    a.sorted_cost = sorted_cost_dev;
    a.perm = perm_dev;
    if (launch_select_sort(static_cast<hipStream_t>(stream), a) != hipSuccess)
        return set_last_error(LMCMA_HIP_ERR_RUNTIME, "rank_noalloc: kernel launch failed");
    return LMCMA_HIP_OK;
}
