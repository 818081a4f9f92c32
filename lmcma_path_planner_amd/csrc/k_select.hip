// Ranking kernel: ascending order of all candidate costs, equal costs in
// candidate order.
//
// Replaces myqsort(lambda, arfitness, arindex, ...) (lmcma_path_planner/src/lmcma.cpp:91-104,
// called at :315).  The reference's comparator looks at the value only and libc's
// merge sort keeps equal values in input order; here every key is the 64-bit
// word (order-preserving image of the float cost << 32) | candidate index, so keys
// are distinct and any correct sort yields that same permutation.
//
// Step 1: each workgroup sorts one tile of 4096 keys in LDS (bitonic network).
// Step 2 (only when there are several tiles): every element adds, over the other
// tiles, the number of keys below its own (binary search) to its position in its
// tile: that sum is its global rank, and it scatters itself there.
#include "device_types.hpp"

namespace lmcma {

namespace {

constexpr int kThreads = 1024;

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

__global__ __launch_bounds__(kThreads) void k_sort_tiles(const float *__restrict__ cost, int lambda,
                                                         uint64_t *__restrict__ keys_out,
                                                         float *__restrict__ sorted_cost,
                                                         int32_t *__restrict__ perm, int single_tile,
                                                         int n /* keys per tile, a power of two */)
{
    __shared__ uint64_t keys[kSortTile];
    const int base = blockIdx.x * kSortTile;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int c = base + i;
        keys[i] = (c < lambda) ? (((uint64_t)order_bits(cost[c]) << 32) | (uint32_t)c)
                               : ~0ull;  // padding sorts last
    }
    __syncthreads();
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < n / 2; i += kThreads) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                const int hi = lo | stride;
                const bool up = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const uint64_t k = keys[i];
        if (single_tile) {
            if (i < lambda) {
                sorted_cost[i] = order_bits_inv((uint32_t)(k >> 32));
                perm[i] = (int32_t)(uint32_t)k;
            }
        } else {
            keys_out[base + i] = k;
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

}  // namespace

int select_num_tiles(int lambda) { return (lambda + kSortTile - 1) / kSortTile; }

hipError_t launch_select(hipStream_t s, const float *cost_all, int lambda, uint64_t *keys_ws,
                         float *sorted_cost, int32_t *perm)
{
    if (lambda <= 0) return hipErrorInvalidValue;
    const int ntiles = select_num_tiles(lambda);
    int n = kSortTile;
    if (ntiles == 1) {  // a single tile only needs the next power of two
        n = 64;
        while (n < lambda) n <<= 1;
    }
    hipLaunchKernelGGL(k_sort_tiles, dim3(ntiles), dim3(kThreads), 0, s, cost_all, lambda, keys_ws,
                       sorted_cost, perm, ntiles == 1 ? 1 : 0, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ntiles == 1) return e;
    const int total = ntiles * kSortTile;
    hipLaunchKernelGGL(k_merge_rank, dim3((total + 255) / 256), dim3(256), 0, s, keys_ws, ntiles,
                       sorted_cost, perm);
    return hipGetLastError();
}

}  // namespace lmcma
