// Probe: how fast can ONE workgroup read data that the previous kernel (all CUs) wrote?
// In-kernel time of the read phase by the 100 MHz counter, for several workgroup sizes,
// access widths and numbers of loads in flight.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coldread_probe.hip -o tools/bin/coldread_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_fill(double *buf, size_t n, double seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        buf[i] = seed + (double)i;
}

// W = doubles per load (1 or 2), U = loads in flight per thread
template <int W, int U>
__global__ __launch_bounds__(1024) void k_read(const double *buf, size_t n, double *out, unsigned long long *stamps)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double acc = 0.0;
    const size_t stride = (size_t)blockDim.x * W;
    for (size_t base = (size_t)threadIdx.x * W; base < n; base += stride * U) {
        double v[U][W];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t at = base + u * stride;
            if (W == 2) {
                const double2 d = at < n ? *reinterpret_cast<const double2 *>(buf + at) : make_double2(0, 0);
                v[u][0] = d.x;
                v[u][W - 1] = d.y;
            } else {
                v[u][0] = at < n ? buf[at] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            for (int w = 0; w < W; ++w) acc += v[u][w];
    }
    out[threadIdx.x] = acc;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[0] = t1 - t0;
}

template <int W, int U>
static void run(const char *what, int threads, size_t bytes, double *buf, double *out, unsigned long long *stamps,
                bool refill)
{
    const size_t n = bytes / 8;
    unsigned long long best = ~0ull, sum = 0, h;
    const int reps = 10;
    for (int i = 0; i < reps; ++i) {
        if (refill) hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, buf, n, 1.0 + i);
        hipLaunchKernelGGL((k_read<W, U>), dim3(1), dim3(threads), 0, 0, buf, n, out, stamps);
        (void)hipMemcpy(&h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        if (h < best) best = h;
        sum += h;
    }
    printf("%-22s %4d threads %2d B/lane %2d in flight %4zu KB : best %6.2f us avg %6.2f us = %6.1f GB/s\n", what,
           threads, W * 8, U, bytes / 1024, best * 0.01, sum * 0.01 / reps, bytes / (sum * 10.0 / reps));
}

int main()
{
    double *buf, *out;
    unsigned long long *stamps;
    (void)hipMalloc(&buf, 1 << 20);
    (void)hipMalloc(&out, 1024 * sizeof(double));
    (void)hipMalloc(&stamps, 16);
    for (int refill = 1; refill >= 0; --refill) {
        const char *what = refill ? "after a writer kernel" : "re-read, no writer";
        for (size_t kb : {16, 64, 256}) {
            const size_t bytes = kb * 1024;
            run<1, 8>(what, 256, bytes, buf, out, stamps, refill);
            run<1, 32>(what, 256, bytes, buf, out, stamps, refill);
            run<2, 8>(what, 256, bytes, buf, out, stamps, refill);
            run<2, 16>(what, 256, bytes, buf, out, stamps, refill);
            run<1, 8>(what, 1024, bytes, buf, out, stamps, refill);
            run<2, 8>(what, 1024, bytes, buf, out, stamps, refill);
            run<2, 16>(what, 1024, bytes, buf, out, stamps, refill);
        }
    }
    return 0;
}
