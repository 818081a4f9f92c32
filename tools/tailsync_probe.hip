// Probe: what does it cost to let the LAST workgroup of a launch consume what the others wrote,
// instead of a second launch?  Producers write one row each and bump a counter; the workgroup
// that sees the final count sums all rows.  Three ways of making the rows visible across XCDs:
//   fence:  plain stores, __threadfence() before the counter, __threadfence() after it
//   sc1:    agent-scope relaxed atomic stores / loads for the rows (write-through, L2 bypass),
//           a workgroup barrier before the counter, no agent-scope fence anywhere
//   two:    the baseline, two launches
// Every iteration uses new values, so a stale read shows as a wrong sum.
// Build: hipcc -O3 --offload-arch=gfx950 tools/tailsync_probe.hip -o tools/bin/tailsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kRowLen = 128;

__device__ __forceinline__ void busy(double &x, int n)
{
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_fused(double *rows, unsigned *counter, double *out, double seed, int work,
                                               unsigned long long *stamps)
{
    __shared__ unsigned s_last;
    __shared__ double s_red[512];
    const int tid = threadIdx.x, nb = gridDim.x;
    double x = seed + blockIdx.x;
    busy(x, work);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tid < kRowLen) {
        double *dst = rows + (size_t)blockIdx.x * kRowLen + tid;
        const double v = seed + blockIdx.x * 0.5 + tid + (x - x);
        if (MODE == 0) *dst = v;
        else __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 0) __threadfence();
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)nb - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) __threadfence();
    // 512 threads: 4 row groups x 128 columns
    const int col = tid & (kRowLen - 1), rg = tid / kRowLen;
    double acc = 0.0;
    for (int b0 = rg; b0 < nb; b0 += 4 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            const double *src = rows + (size_t)(b < nb ? b : 0) * kRowLen + col;
            v[u] = MODE == 0 ? *src : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b >= nb) v[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    s_red[tid] = acc;
    __syncthreads();
    if (tid < kRowLen) out[tid] = s_red[tid] + s_red[tid + 128] + s_red[tid + 256] + s_red[tid + 384];
    if (tid == 0) {
        *counter = 0u;  // (the next launch starts after this one has drained)
        stamps[0] = t1 - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - t1;
    }
}

// The shape the library would use: a dedicated extra workgroup (the last of the grid) waits for the
// counter; producers store through (agent-scope atomic stores), the waiting workgroup takes one
// agent-scope acquire fence and then reads with ordinary 16-byte loads.
__global__ __launch_bounds__(512) void k_tail(double *rows, unsigned *counter, double *out, double seed, int work,
                                              unsigned long long *stamps)
{
    __shared__ double s_red[512];
    const int tid = threadIdx.x, nb = gridDim.x - 1;
    if ((int)blockIdx.x < nb) {
        double x = seed + blockIdx.x;
        busy(x, work);
        if (tid < kRowLen)
            __hip_atomic_store(rows + (size_t)blockIdx.x * kRowLen + tid, seed + blockIdx.x * 0.5 + tid + (x - x),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nb && ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // 512 threads: 8 row groups x 64 column pairs
    const int cp = tid & 63, rg = tid >> 6;
    double a0 = 0.0, a1 = 0.0;
    for (int b0 = rg; b0 < nb; b0 += 8 * 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 8 * u;
            v[u] = make_double2(0.0, 0.0);
            if (b < nb) v[u] = *reinterpret_cast<const double2 *>(rows + (size_t)b * kRowLen + 2 * cp);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += v[u].x; a1 += v[u].y; }
    }
    __syncthreads();
    s_red[tid] = a0;
    __syncthreads();
    double t = 0.0;
    if (tid < 64) for (int r = 0; r < 8; ++r) t += s_red[r * 64 + tid];
    __syncthreads();
    s_red[tid] = a1;
    __syncthreads();
    if (tid < 64) {
        double u2 = 0.0;
        for (int r = 0; r < 8; ++r) u2 += s_red[r * 64 + tid];
        out[2 * tid] = t;
        out[2 * tid + 1] = u2;
    }
    if (tid == 0) {
        *counter = 0u;
        stamps[0] = t1 - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - t1;
    }
}

__global__ __launch_bounds__(512) void k_produce(double *rows, double seed, int work)
{
    double x = seed + blockIdx.x;
    busy(x, work);
    if (threadIdx.x < kRowLen)
        rows[(size_t)blockIdx.x * kRowLen + threadIdx.x] = seed + blockIdx.x * 0.5 + threadIdx.x + (x - x);
}

__global__ __launch_bounds__(512) void k_consume(const double *rows, int nb, double *out)
{
    __shared__ double s_red[512];
    const int tid = threadIdx.x, col = tid & (kRowLen - 1), rg = tid / kRowLen;
    double acc = 0.0;
    for (int b0 = rg; b0 < nb; b0 += 4 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            v[u] = b < nb ? rows[(size_t)b * kRowLen + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    s_red[tid] = acc;
    __syncthreads();
    if (tid < kRowLen) out[tid] = s_red[tid] + s_red[tid + 128] + s_red[tid + 256] + s_red[tid + 384];
}

#define CK(e)                                                                 \
    do {                                                                      \
        hipError_t _e = (e);                                                  \
        if (_e != hipSuccess) {                                               \
            printf("%s: %s\n", #e, hipGetErrorString(_e));                    \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static double expect(int nb, double seed, int col)
{
    // the kernels add in a fixed order; the check allows rounding
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += seed + b * 0.5 + col;
    return s;
}

int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 128, iters = 300, work = argc > 2 ? atoi(argv[2]) : 2000;
    double *rows, *out;
    unsigned *counter;
    unsigned long long *stamps;
    CK(hipMalloc(&rows, (size_t)nb * kRowLen * sizeof(double)));
    CK(hipMalloc(&out, kRowLen * sizeof(double)));
    CK(hipMalloc(&counter, sizeof(unsigned)));
    CK(hipMalloc(&stamps, 2 * sizeof(unsigned long long)));
    CK(hipMemset(counter, 0, sizeof(unsigned)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double host[kRowLen];
    const char *names[4] = {"fence", "sc1", "two launches", "waiting tail"};
    for (int mode = 0; mode < 4; ++mode) {
        int bad = 0;
        float ms = 0.f;
        unsigned long long st[2] = {0, 0};
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < iters; ++it) {
                const double seed = 1.0 + it + 1000.0 * rep;
                if (mode == 0) hipLaunchKernelGGL(k_fused<0>, dim3(nb), dim3(512), 0, s, rows, counter, out, seed, work, stamps);
                else if (mode == 1) hipLaunchKernelGGL(k_fused<1>, dim3(nb), dim3(512), 0, s, rows, counter, out, seed, work, stamps);
                else if (mode == 3) hipLaunchKernelGGL(k_tail, dim3(nb + 1), dim3(512), 0, s, rows, counter, out, seed, work, stamps);
                else {
                    hipLaunchKernelGGL(k_produce, dim3(nb), dim3(512), 0, s, rows, seed, work);
                    hipLaunchKernelGGL(k_consume, dim3(1), dim3(512), 0, s, rows, nb, out);
                }
                if (rep == 0 && it % 37 == 0) {  // check some iterations of the warm-up pass (the copy drains the stream)
                    CK(hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    for (int c = 0; c < kRowLen; ++c) {
                        const double w = expect(nb, seed, c);
                        if (!(host[c] > w - 1e-6 * w && host[c] < w + 1e-6 * w)) ++bad;
                    }
                }
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        if (mode != 2) CK(hipMemcpy(st, stamps, sizeof(st), hipMemcpyDeviceToHost));
        printf("%-13s nb %d: %.2f us per iteration, wrong values %d", names[mode], nb, ms * 1000.f / iters, bad);
        if (mode != 2) printf(", last workgroup: store->last (or wait) %.2f us, fence+read+sum %.2f us", st[0] * 0.01, st[1] * 0.01);
        printf("\n");
    }
    return 0;
}
