// Clearance map on the device: exact Euclidean distance transform of an occupancy grid.
//
// The reference builds its EDT_Matrix with dynamicEDT3D (sample_based_optimisation_based_
// path_planner.cpp:304-307, 319: getDistance of every cell, in cells) and scores a state by
// 1 / clearance (:655-668).  dynamicEDT3D's values are the exact Euclidean distances to the
// nearest occupied cell; squared, they are integers, so any exact algorithm gives the same
// bits.  Here: three separable passes over squared distances held as uint32,
//     g(a) = min over a' of f(a') + (a - a')^2      along x, then y, then z.
// Along x the input is binary and a line needs two scans; along y and z every output searches
// outwards from its own position until k^2 can no longer beat the best value found.  The y
// and z passes stage tiles of 16 to 64 neighbouring columns in LDS, so that every global
// access is a contiguous run along x.
#include <utility>

#include "device_types.hpp"

namespace lmcma {

namespace {

constexpr uint32_t kEdtFar = 0x3fffffffu;  // no occupied cell on the lines searched so far

// along x: one wave per line.  The input is binary, so the nearest occupied cell of the line
// is the nearer of the last one at or before x and the first one at or after x: a prefix
// maximum and a suffix minimum over the line, 64 cells per step with a carry between steps.
__device__ __forceinline__ int wave_prefix_max(int v, int lane)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(v, s, 64);
        if (lane >= s) v = max(v, o);
    }
    return v;
}

__device__ __forceinline__ int wave_suffix_min(int v, int lane)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_down(v, s, 64);
        if (lane + s < 64) v = min(v, o);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_edt_x(const uint8_t *__restrict__ occ, uint32_t *__restrict__ d2,
                                               int nx, long lines)
{
    extern __shared__ uint32_t s_line[];  // [4][64 * chunks]: last occupied cell at or before x
    constexpr int kNone = 1 << 20;        // farther than any line is long
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = (nx + 63) >> 6;
    int *left = reinterpret_cast<int *>(s_line) + (size_t)wave * chunks * 64;
    for (long line = (long)blockIdx.x * 4 + wave; line < lines; line += (long)gridDim.x * 4) {
        const uint8_t *src = occ + line * nx;
        int carry = -kNone;
        for (int c = 0; c < chunks; ++c) {
            const int x = c * 64 + lane;
            int v = (x < nx && src[x]) ? x : -kNone;
            v = max(wave_prefix_max(v, lane), carry);
            left[x] = v;  // read back by the same lane only
            carry = __shfl(v, 63, 64);
        }
        carry = kNone;
        for (int c = chunks - 1; c >= 0; --c) {
            const int x = c * 64 + lane;
            int v = (x < nx && src[x]) ? x : kNone;
            v = min(wave_suffix_min(v, lane), carry);
            carry = __shfl(v, 0, 64);
            if (x < nx) {
                const int dl = x - left[x], dr = v - x;
                const int d = dl < dr ? dl : dr;
                d2[line * nx + x] = d >= kNone - 2048 ? kEdtFar : (uint32_t)(d * d);
            }
        }
    }
}

// squared distance -> distance (float) and the state cost 1 / max(distance, d_min); a grid
// without any occupied cell reads 1e6 cells everywhere
struct EdtOut {
    float *dist, *cost;  // either may be null
    double d_min;
};

__device__ __forceinline__ void edt_store(const EdtOut &f, long i, uint32_t v)
{
    const float d = v >= kEdtFar ? 1.0e6f : (float)sqrt((double)v);
    if (f.dist) f.dist[i] = d;
    if (f.cost) f.cost[i] = (float)(1.0 / fmax((double)d, f.d_min));
}

// along an axis with a stride: element (o, a, x) lives at o * outer_stride + a * axis_stride + x.
// A workgroup stages TX neighbouring columns of one o (all n positions of the axis) in LDS and
// computes one of `segs` interleaved shares of the positions (a small grid has too few column
// tiles to fill the chip otherwise; inputs and outputs are different arrays for that reason).
// Every output searches outwards from its own position, eight steps (sixteen LDS reads in
// flight) at a time, and stops once k^2 can no longer beat the best value found: on a map
// with obstacles every few cells that is one or two rounds, on an empty one n / 8.
// LAST: the pass writes distances / costs (fin) instead of squared distances.
template <int TX, bool LAST>
__global__ __launch_bounds__(256) void k_edt_axis(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int nx,
                                                  int n, long axis_stride, long outer_stride, int n_outer, int segs,
                                                  const EdtOut fin)
{
    extern __shared__ uint32_t s_tile[];  // [n][TX]
    constexpr int AY = 256 / TX;          // positions of the axis handled at once
    constexpr int KB = 8;                 // steps per round
    const int tx = threadIdx.x % TX, ta = threadIdx.x / TX;
    const int xtiles = (nx + TX - 1) / TX;
    const long work_items = (long)xtiles * n_outer * segs;
    for (long work = blockIdx.x; work < work_items; work += gridDim.x) {
        const int seg = (int)(work % segs);
        const long tile = work / segs;
        const int o = (int)(tile / xtiles), x = (int)(tile % xtiles) * TX + tx;
        const bool live = x < nx;
        const long base = (long)o * outer_stride + x;
        __syncthreads();  // the previous tile has been searched
        for (int a0 = ta; a0 < n; a0 += 8 * AY) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = a0 + u * AY;
                v[u] = (live && a < n) ? in[base + (long)a * axis_stride] : kEdtFar;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = a0 + u * AY;
                if (a < n) s_tile[a * TX + tx] = v[u];
            }
        }
        __syncthreads();
        if (!live) continue;
        for (int a = ta + seg * AY; a < n; a += AY * segs) {
            uint32_t best = s_tile[a * TX + tx];
            for (int k0 = 1; k0 < n && (uint32_t)(k0 * k0) < best; k0 += KB) {
                uint32_t lo[KB], hi[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int d = a - (k0 + u), e = a + (k0 + u);
                    lo[u] = s_tile[(d < 0 ? 0 : d) * TX + tx];
                    hi[u] = s_tile[(e >= n ? n - 1 : e) * TX + tx];
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k = k0 + u;
                    const uint32_t kk = (uint32_t)(k * k);
                    // positions past the ends were read at a clamped index: dropped here
                    if (a - k >= 0) best = min(best, lo[u] + kk);
                    if (a + k < n) best = min(best, hi[u] + kk);
                }
            }
            if (LAST) edt_store(fin, base + (long)a * axis_stride, best);
            else out[base + (long)a * axis_stride] = best;
        }
    }
}

// the same conversion as a pass of its own (a 1-D grid has no y or z pass to carry it)
__global__ __launch_bounds__(256) void k_edt_finish(const uint32_t *__restrict__ d2, long cells, const EdtOut fin)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < cells; i += (long)gridDim.x * 256) edt_store(fin, i, d2[i]);
}

template <int TX, bool LAST>
hipError_t launch_axis(hipStream_t s, const uint32_t *in, uint32_t *out, int nx, int n, long axis_stride,
                       long outer_stride, int n_outer, const EdtOut &fin)
{
    const size_t lds = (size_t)n * TX * sizeof(uint32_t);
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_edt_axis<TX, LAST>), lds);
        if (e != hipSuccess) return e;
    }
    const long tiles = (long)((nx + TX - 1) / TX) * n_outer;
    // about four workgroups per CU, each with at least one round of positions
    int segs = (int)((1024 + tiles - 1) / tiles);
    const int max_segs = (n + (256 / TX) - 1) / (256 / TX);
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    const long work = tiles * segs;
    const int blocks = (int)(work < 16384 ? work : 16384);
    hipLaunchKernelGGL((k_edt_axis<TX, LAST>), dim3(blocks), dim3(256), lds, s, in, out, nx, n, axis_stride, outer_stride,
                       n_outer, segs, fin);
    return hipGetLastError();
}

// columns per workgroup: as many as keep the tile within 32 KiB (four workgroups per CU hide the
// LDS latency of a round better than two with twice the columns), at least 16 (64-byte runs)
template <bool LAST>
hipError_t edt_axis(hipStream_t s, const uint32_t *in, uint32_t *out, int nx, int n, long axis_stride,
                    long outer_stride, int n_outer, const EdtOut &fin)
{
    if (n <= 128) return launch_axis<64, LAST>(s, in, out, nx, n, axis_stride, outer_stride, n_outer, fin);
    if (n <= 256) return launch_axis<32, LAST>(s, in, out, nx, n, axis_stride, outer_stride, n_outer, fin);
    return launch_axis<16, LAST>(s, in, out, nx, n, axis_stride, outer_stride, n_outer, fin);
}

}  // namespace

// occ: [nz][ny][nx] bytes on the device (non-zero = occupied); d2: [2][cells] scratch (the
// passes alternate between the halves); dist, cost: [cells] floats or null.  Extents up to
// 1024 per axis.
hipError_t launch_edt(hipStream_t s, const uint8_t *occ, int nx, int ny, int nz, double d_min, uint32_t *d2,
                      float *dist, float *cost)
{
    if (nx < 1 || ny < 1 || nz < 1 || nx > 1024 || ny > 1024 || nz > 1024 || !(d_min > 0.0)) return hipErrorInvalidValue;
    const long cells = (long)nx * ny * nz, lines = (long)ny * nz;
    {
        const long groups = (lines + 3) / 4;
        const int blocks = (int)(groups < 8192 ? groups : 8192);
        const size_t lds = (size_t)4 * ((nx + 63) / 64) * 64 * sizeof(uint32_t);
        hipLaunchKernelGGL(k_edt_x, dim3(blocks), dim3(256), lds, s, occ, d2, nx, lines);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    uint32_t *cur = d2, *other = d2 + cells;
    const EdtOut fin{dist, cost, d_min};
    // the last pass along an axis writes the distances and costs itself
    if (ny > 1) {
        hipError_t e = nz > 1 ? edt_axis<false>(s, cur, other, nx, ny, nx, (long)nx * ny, nz, fin)
                              : edt_axis<true>(s, cur, other, nx, ny, nx, (long)nx * ny, nz, fin);
        if (e != hipSuccess || nz == 1) return e;
        std::swap(cur, other);
    }
    if (nz > 1) return edt_axis<true>(s, cur, other, nx, nz, (long)nx * ny, nx, ny, fin);
    const long groups = (cells + 255) / 256;
    hipLaunchKernelGGL(k_edt_finish, dim3((int)(groups < 16384 ? groups : 16384)), dim3(256), 0, s, cur, cells, fin);
    return hipGetLastError();
}

}  // namespace lmcma
