// Clearance map on the device: exact Euclidean distance transform of an occupancy grid.
//
// The reference builds its EDT_Matrix with dynamicEDT3D (sample_based_optimisation_based_
// path_planner.cpp:304-307, 319: getDistance of every cell, in cells) and scores a state by
// 1 / clearance (:655-668).  dynamicEDT3D's values are the exact Euclidean distances to the
// nearest occupied cell; squared, they are integers, so any exact algorithm gives the same
// bits.  Here: three separable passes over squared distances held as uint32,
//     g(a) = min over a' of f(a') + (a - a')^2      along x, then y, then z.
// Along x the input is binary and a line needs two scans; along y and z a lane walks a
// column with an integer lower-envelope scan.  The y and z passes stage tiles of 16 to 64
// neighbouring columns in LDS, so that every global access is a contiguous run along x.
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

// along an axis with a stride: element (o, a, x) lives at o * outer_stride + a * axis_stride + x.
// One wave stages TX neighbouring columns of one o (all n positions of the axis) in LDS and
// one lane walks one column with the integer lower-envelope scan of Meijster, Roerdink and
// Hesselink: a forward sweep keeps the stack of parabolas (s = their apex, t = the position
// from which each is the lowest), a backward sweep reads the minimum off it.  Every step is
// exact integer arithmetic: with kEdtFar < 2^30 and n <= 1024 nothing leaves 31 bits.
// O(n) per column whatever the map looks like.
template <int TX>
__global__ __launch_bounds__(64) void k_edt_axis(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int nx,
                                                 int n, long axis_stride, long outer_stride, int n_outer)
{
    extern __shared__ uint32_t s_mem[];
    int *g = reinterpret_cast<int *>(s_mem);        // [n][TX] the column values
    uint32_t *stack = s_mem + (size_t)n * TX;       // [n][TX] s | t << 16
    constexpr int AY = 64 / TX;
    const int lane = threadIdx.x, tx = lane % TX, ta = lane / TX;
    const int xtiles = (nx + TX - 1) / TX;
    const long tiles = (long)xtiles * n_outer;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int o = (int)(tile / xtiles), x = (int)(tile % xtiles) * TX + tx;
        const bool live = x < nx;
        const long base = (long)o * outer_stride + x;
        __syncthreads();  // the previous tile's columns are finished
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
                if (a < n) g[a * TX + tx] = (int)v[u];
            }
        }
        __syncthreads();
        if (lane < TX && live) {
            // F(p, i) = (p - i)^2 + g(i): the parabola with apex i, at position p
            int q = 0;
            stack[tx] = 0u;  // s = 0, t = 0
            int sq = 0, tq = 0, gq = g[tx];  // top of the stack, and g at its apex
            for (int u = 1; u < n; ++u) {
                const int gu = g[u * TX + tx];
                // parabolas that u beats already where they start are never the lowest
                while (q >= 0 && (tq - sq) * (tq - sq) + gq > (tq - u) * (tq - u) + gu) {
                    --q;
                    if (q >= 0) {
                        const uint32_t e = stack[q * TX + tx];
                        sq = (int)(e & 0xffffu);
                        tq = (int)(e >> 16);
                        gq = g[sq * TX + tx];
                    }
                }
                if (q < 0) {
                    q = 0;
                    sq = u; tq = 0; gq = gu;
                    stack[tx] = (uint32_t)u;
                } else {
                    // first position where u is lower than the top: 1 + floor of the crossing
                    const int w = 1 + (u * u - sq * sq + gu - gq) / (2 * (u - sq));
                    if (w < n) {
                        ++q;
                        sq = u; tq = w; gq = gu;
                        stack[q * TX + tx] = (uint32_t)u | ((uint32_t)w << 16);
                    }
                }
            }
            for (int u = n - 1; u >= 0; --u) {
                const int d = (u - sq) * (u - sq) + gq;
                out[base + (long)u * axis_stride] = d >= (int)kEdtFar ? kEdtFar : (uint32_t)d;
                if (u == tq && q > 0) {
                    --q;
                    const uint32_t e = stack[q * TX + tx];
                    sq = (int)(e & 0xffffu);
                    tq = (int)(e >> 16);
                    gq = g[sq * TX + tx];
                }
            }
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
hipError_t launch_axis(hipStream_t s, const uint32_t *in, uint32_t *out, int nx, int n, long axis_stride,
                       long outer_stride, int n_outer)
{
    const size_t lds = 2 * (size_t)n * TX * sizeof(uint32_t);
    static size_t configured = 48 * 1024;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_edt_axis<TX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    const long tiles = (long)((nx + TX - 1) / TX) * n_outer;
    const int blocks = (int)(tiles < 16384 ? tiles : 16384);
    hipLaunchKernelGGL(k_edt_axis<TX>, dim3(blocks), dim3(64), lds, s, in, out, nx, n, axis_stride, outer_stride, n_outer);
    return hipGetLastError();
}

// columns per wave: as many as leave the tile and the stacks (8 n bytes per column) within 128 KiB
hipError_t edt_axis(hipStream_t s, const uint32_t *in, uint32_t *out, int nx, int n, long axis_stride,
                    long outer_stride, int n_outer)
{
    if (n <= 256) return launch_axis<64>(s, in, out, nx, n, axis_stride, outer_stride, n_outer);
    if (n <= 512) return launch_axis<32>(s, in, out, nx, n, axis_stride, outer_stride, n_outer);
    return launch_axis<16>(s, in, out, nx, n, axis_stride, outer_stride, n_outer);
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
    if (ny > 1) {
        hipError_t e = edt_axis(s, cur, other, nx, ny, nx, (long)nx * ny, nz);
        if (e != hipSuccess) return e;
        std::swap(cur, other);
    }
    if (nz > 1) {
        hipError_t e = edt_axis(s, cur, other, nx, nz, (long)nx * ny, nx, ny);
        if (e != hipSuccess) return e;
        std::swap(cur, other);
    }
    const long groups = (cells + 255) / 256;
    hipLaunchKernelGGL(k_edt_finish, dim3((int)(groups < 16384 ? groups : 16384)), dim3(256), 0, s, cur, cells, d_min,
                       dist, cost);
    return hipGetLastError();
}

}  // namespace lmcma
