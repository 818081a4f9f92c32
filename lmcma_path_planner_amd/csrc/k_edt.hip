// Clearance map on the device: exact Euclidean distance transform of an occupancy grid.
//
// The reference builds its EDT_Matrix with dynamicEDT3D (sample_based_optimisation_based_
// path_planner.cpp:304-307, 319: getDistance of every cell, in cells) and scores a state by
// 1 / clearance (:655-668).  dynamicEDT3D's values are the exact Euclidean distances to the
// nearest occupied cell; squared, they are integers, so any exact algorithm gives the same
// bits.  Here: three separable passes over squared distances held as uint32,
//     g(a) = min over a' of f(a') + (a - a')^2      along x, then y, then z,
// each output searching outwards from its own position and stopping as soon as k^2 can no
// longer beat the best value found (at 15 % occupancy that is a handful of steps, not the
// line length).  A pass stages its lines in LDS: x lines whole, y and z lines as tiles of
// 32 or 64 neighbouring columns so that every global access is a contiguous run along x.
#include "device_types.hpp"

namespace lmcma {

namespace {

constexpr uint32_t kEdtFar = 0x3fffffffu;  // no occupied cell on the lines searched so far

// along x: one wave per line, the line in LDS as 0 / kEdtFar
__global__ __launch_bounds__(256) void k_edt_x(const uint8_t *__restrict__ occ, uint32_t *__restrict__ d2,
                                               int nx, long lines)
{
    extern __shared__ uint32_t s_line[];  // [4][nx]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *f = s_line + (size_t)wave * nx;
    for (long line0 = (long)blockIdx.x * 4; line0 < lines; line0 += (long)gridDim.x * 4) {
        const long line = line0 + wave;
        const bool live = line < lines;
        __syncthreads();  // the previous lines have been searched
        if (live)
            for (int x = lane; x < nx; x += 64) f[x] = occ[line * nx + x] ? 0u : kEdtFar;
        __syncthreads();
        if (!live) continue;
        for (int x = lane; x < nx; x += 64) {
            uint32_t best = f[x];
            // the first occupied cell met going outwards is the nearest
            for (int k = 1; best == kEdtFar && (x - k >= 0 || x + k < nx); ++k) {
                const bool hit = (x - k >= 0 && f[x - k] == 0u) || (x + k < nx && f[x + k] == 0u);
                if (hit) best = (uint32_t)(k * k);
            }
            d2[line * nx + x] = best;
        }
    }
}

// along an axis with a stride: element (o, a, x) lives at o * outer_stride + a * axis_stride + x.
// A workgroup owns TX neighbouring columns of one o, all n positions of the axis.
template <int TX>
__global__ __launch_bounds__(256) void k_edt_axis(uint32_t *__restrict__ d2, int nx, int n, long axis_stride,
                                                  long outer_stride, int n_outer)
{
    extern __shared__ uint32_t s_tile[];  // [n][TX]
    constexpr int AY = 256 / TX;          // positions of the axis handled at once
    const int tx = threadIdx.x % TX, ta = threadIdx.x / TX;
    const int xtiles = (nx + TX - 1) / TX;
    const long tiles = (long)xtiles * n_outer;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int o = (int)(tile / xtiles), x = (int)(tile % xtiles) * TX + tx;
        const bool live = x < nx;
        uint32_t *base = d2 + (long)o * outer_stride + x;
        __syncthreads();  // the previous tile has been searched
        for (int a = ta; a < n; a += AY) s_tile[a * TX + tx] = live ? base[(long)a * axis_stride] : kEdtFar;
        __syncthreads();
        if (!live) continue;
        for (int a = ta; a < n; a += AY) {
            uint32_t best = s_tile[a * TX + tx];
            for (int k = 1; (uint32_t)(k * k) < best && (a - k >= 0 || a + k < n); ++k) {
                const uint32_t kk = (uint32_t)(k * k);
                if (a - k >= 0) best = min(best, s_tile[(a - k) * TX + tx] + kk);
                if (a + k < n) best = min(best, s_tile[(a + k) * TX + tx] + kk);
            }
            base[(long)a * axis_stride] = best;
        }
    }
}

// squared distance -> distance (float) and the state cost 1 / max(distance, d_min); a grid
// without any occupied cell reads 1e6 cells everywhere
__global__ __launch_bounds__(256) void k_edt_finish(const uint32_t *__restrict__ d2, long cells, double d_min,
                                                    float *__restrict__ dist, float *__restrict__ cost)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < cells; i += (long)gridDim.x * 256) {
        const uint32_t v = d2[i];
        const float d = v >= kEdtFar ? 1.0e6f : (float)sqrt((double)v);
        if (dist) dist[i] = d;
        if (cost) cost[i] = (float)(1.0 / fmax((double)d, d_min));
    }
}

template <int TX>
hipError_t launch_axis(hipStream_t s, uint32_t *d2, int nx, int n, long axis_stride, long outer_stride, int n_outer)
{
    const size_t lds = (size_t)n * TX * sizeof(uint32_t);
    static size_t configured = 48 * 1024;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edt_axis<TX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    const long tiles = (long)((nx + TX - 1) / TX) * n_outer;
    const int blocks = (int)(tiles < 8192 ? tiles : 8192);
    hipLaunchKernelGGL(k_edt_axis<TX>, dim3(blocks), dim3(256), lds, s, d2, nx, n, axis_stride, outer_stride, n_outer);
    return hipGetLastError();
}

hipError_t edt_axis(hipStream_t s, uint32_t *d2, int nx, int n, long axis_stride, long outer_stride, int n_outer)
{
    if (n <= 512) return launch_axis<64>(s, d2, nx, n, axis_stride, outer_stride, n_outer);
    return launch_axis<32>(s, d2, nx, n, axis_stride, outer_stride, n_outer);
}

}  // namespace

// occ: [nz][ny][nx] bytes on the device (non-zero = occupied); d2: [cells] scratch; dist, cost:
// [cells] floats or null.  Extents up to 1024 per axis.
hipError_t launch_edt(hipStream_t s, const uint8_t *occ, int nx, int ny, int nz, double d_min, uint32_t *d2,
                      float *dist, float *cost)
{
    if (nx < 1 || ny < 1 || nz < 1 || nx > 1024 || ny > 1024 || nz > 1024 || !(d_min > 0.0)) return hipErrorInvalidValue;
    const long cells = (long)nx * ny * nz, lines = (long)ny * nz;
    {
        const long groups = (lines + 3) / 4;
        const int blocks = (int)(groups < 8192 ? groups : 8192);
        hipLaunchKernelGGL(k_edt_x, dim3(blocks), dim3(256), (size_t)4 * nx * sizeof(uint32_t), s, occ, d2, nx, lines);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (ny > 1) {
        hipError_t e = edt_axis(s, d2, nx, ny, nx, (long)nx * ny, nz);
        if (e != hipSuccess) return e;
    }
    if (nz > 1) {
        hipError_t e = edt_axis(s, d2, nx, nz, (long)nx * ny, nx, ny);
        if (e != hipSuccess) return e;
    }
    const long groups = (cells + 255) / 256;
    hipLaunchKernelGGL(k_edt_finish, dim3((int)(groups < 16384 ? groups : 16384)), dim3(256), 0, s, d2, cells, d_min,
                       dist, cost);
    return hipGetLastError();
}

}  // namespace lmcma
