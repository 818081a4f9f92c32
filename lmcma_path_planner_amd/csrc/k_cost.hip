// Trajectory cost kernel: one cost per candidate row.
//
// Stands in for the caller's evaluation loop between getNextParameterVector and
// setEvaluationFeedback (lmcma_path_planner/src/example_lmcma.cpp:51-54).  The
// reference never connects a trajectory cost to LM-CMA; the terms are assembled
// from its planning ingredients (S/ = sample_based_optimisation_based_path_planner/src/
// sample_based_optimisation_based_path_planner.cpp, L/ = lmcma_path_planner/src/lmcma.cpp):
//   cell lookup with round-half-even, (row = y, col = x)      S/:591-603
//   state cost read from the looked-up cell                   S/:605-628, 655-668
//   path length                                               S/:634-639
//   weighted sum of the objectives                            S/:677-690
//   5-tap acceleration rule, truncated at the path ends       L/:763, 823-832
//   the three terms summed left to right                      L/:186-188
//
// Path: node 0 = start, nodes 1..W = the row's waypoints, node W+1 = goal.
//   J_len    = sum over the W+1 segments of their Euclidean length
//   J_grid   = sum over segments of 1/2 (c(a) + c(b)) * length   (trapezoid)
//   J_smooth = sum over waypoints and dimensions of acceleration^2
//   cost     = w_len J_len + w_grid J_grid + w_smooth J_smooth
//
// Work split: G lanes own one candidate and lane l owns waypoints l, l+G, ...;
// a waypoint's lane sums 1/2 c(node) (len_prev + len_next) — the trapezoid sum
// regrouped per node, so every cell is looked up once — plus the endpoint nodes'
// shares at the first and last waypoint.  Rows are staged through LDS with
// coalesced 16-byte loads so the +-2 neighbour reads are conflict-free LDS
// reads; a bit-packed occupancy grid can be staged beside them.
#include "device_funcs.hpp"

namespace lmcma {

namespace {

template <int G, int D, bool GRID_LDS>
__global__ __launch_bounds__(256) void k_cost(const CostParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int C = 256 / G;  // candidates per tile
    const int W = p.W;
    const int ld = p.ld;
    float *srows = smem;  // [C][ld]
    const uint32_t *bits;
    if (GRID_LDS) {
        uint32_t *sbits = reinterpret_cast<uint32_t *>(smem + C * ld);
        const uint4 *src = reinterpret_cast<const uint4 *>(p.grid);
        uint4 *dst = reinterpret_cast<uint4 *>(sbits);
        for (int i = threadIdx.x; i < (p.grid_words >> 2); i += 256) dst[i] = src[i];
        bits = sbits;
    } else {
        bits = reinterpret_cast<const uint32_t *>(p.grid);
    }

    const int g = threadIdx.x / G;
    const int lane = threadIdx.x % G;
    const int ntiles = (p.n_rows + C - 1) / C;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * C;
        const int nrows = min(C, p.n_rows - row0);
        __syncthreads();  // previous tile fully consumed (and the grid copy issued)
        {
            const float4 *src = reinterpret_cast<const float4 *>(p.rows + (size_t)row0 * ld);
            float4 *dst = reinterpret_cast<float4 *>(srows);
            const int n4 = (nrows * ld) >> 2;
            for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
        }
        __syncthreads();

        float j_len = 0.f, j_grid = 0.f, j_smooth = 0.f;
        if (g < nrows) {
            const float *row = srows + g * ld;
            for (int k = lane; k < W; k += G) waypoint_terms<D>(p, bits, row, k, j_len, j_grid, j_smooth);
        }
        j_len = group_sum<G>(j_len);
        j_grid = group_sum<G>(j_grid);
        j_smooth = group_sum<G>(j_smooth);
        if (g < nrows && lane == 0) {
            const float total = total_cost(p, j_len, j_grid, j_smooth);
            p.cost[row0 + g] = total;
            if (p.terms) {
                float *t = p.terms + 3 * (size_t)(row0 + g);
                t[0] = j_len; t[1] = j_grid; t[2] = j_smooth;
            }
        }
    }
}

template <int G, int D>
hipError_t launch_gd(hipStream_t s, const CostParams &p, bool lds_grid)
{
    constexpr int C = 256 / G;
    const int ntiles = (p.n_rows + C - 1) / C;
    size_t lds = (size_t)C * p.ld * sizeof(float);
    if (lds_grid) {
        lds += (size_t)p.grid_words * sizeof(uint32_t);
        // one block per CU re-uses its staged grid over several row tiles
        const int blocks = min(ntiles, 256);
        // raise the kernel's dynamic-LDS limit once per size (not a stream operation,
        // so it is kept out of the steady-state launch path and of graph captures)
        {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_cost<G, D, true>), lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((k_cost<G, D, true>), dim3(blocks), dim3(256), lds, s, p);
    } else {
        const int blocks = min(ntiles, 2048);
        hipLaunchKernelGGL((k_cost<G, D, false>), dim3(blocks), dim3(256), lds, s, p);
    }
    return hipGetLastError();
}

template <int D>
hipError_t launch_d(hipStream_t s, const CostParams &p, bool lds_grid)
{
    // lanes per candidate: the smallest group that leaves the fewest idle lanes
    const int W = p.W;
    int G;
    if (W % 64 == 0 || W > 96) G = 64;
    else if (W % 32 == 0) G = 32;
    else if (W % 16 == 0 || W < 16) G = 16;
    else if (W <= 32) G = 32;
    else G = 64;
    switch (G) {
        case 16: return launch_gd<16, D>(s, p, lds_grid);
        case 32: return launch_gd<32, D>(s, p, lds_grid);
        default: return launch_gd<64, D>(s, p, lds_grid);
    }
}

}  // namespace

hipError_t launch_cost(hipStream_t s, const CostParams &p, int dims, bool lds_grid)
{
    if (p.n_rows <= 0) return hipSuccess;
    if (p.W < 1 || (p.ld & 3) || dims * p.W > p.ld) return hipErrorInvalidValue;
    if (lds_grid && (p.grid_type != LMCMA_GRID_BITS || (p.grid_words & 3))) return hipErrorInvalidValue;
    switch (dims) {
        case 1: return launch_d<1>(s, p, lds_grid);
        case 2: return launch_d<2>(s, p, lds_grid);
        case 3: return launch_d<3>(s, p, lds_grid);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lmcma
