// Probe: cost of executing straight-line code for the first time in a kernel
// (instruction fetch from L2/HBM) against the same code executed a second time.
// Build: hipcc -O3 --offload-arch=gfx950 tools/icache_probe.hip -o tools/bin/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int REP>
__global__ __launch_bounds__(1024) void k_probe(double *out, unsigned long long *stamps, double seed)
{
    double x = seed + threadIdx.x, y = seed * 0.5;
    unsigned long long t[3];
    t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            x = __builtin_fma(x, y, 1.0 + r);  // 8-byte VOP3 with a distinct literal -> no folding
            y = __builtin_fma(y, x, 0.5);
        }
        __builtin_amdgcn_s_waitcnt(0);
        t[pass + 1] = __builtin_amdgcn_s_memtime();
    }
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) {
        stamps[0] = t[1] - t[0];
        stamps[1] = t[2] - t[1];
    }
}

template <int REP>
static void run(int threads, double *out, unsigned long long *stamps)
{
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_probe<REP>));
    unsigned long long best[2] = {~0ull, ~0ull}, h[2];
    unsigned long long sum[2] = {0, 0};
    const int reps = 20;
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(k_probe<REP>, dim3(1), dim3(threads), 0, 0, out, stamps, 1.0 + i);
        (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int q = 0; q < 2; ++q) {
            if (h[q] < best[q]) best[q] = h[q];
            sum[q] += h[q];
        }
    }
    printf("REP %5d (~%6d B of fma) threads %4d : first pass %7llu (avg %7llu)  second pass %7llu (avg %7llu) ticks\n",
           REP, REP * 2 * 8, threads, best[0], sum[0] / reps, best[1], sum[1] / reps);
}

int main()
{
    double *out;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 1024 * sizeof(double));
    (void)hipMalloc(&stamps, 16);
    for (int threads : {64, 1024}) {
        run<32>(threads, out, stamps);
        run<128>(threads, out, stamps);
        run<512>(threads, out, stamps);
        run<1024>(threads, out, stamps);
        run<2048>(threads, out, stamps);
    }
    // tick rate of s_memtime against the 100 MHz realtime counter is printed by tools/diag_update.py
    (void)hipFree(out);
    (void)hipFree(stamps);
    return 0;
}
