// Elite recombination and the on-device LM-CMA update (fp64).
//
// Replaces LMCMA::update (lmcma_path_planner/src/lmcma.cpp:313-424) and invAz
// (:449-463); the citations L/:a-b below are lines of that file.
#include "device_types.hpp"

namespace lmcma {

namespace {

// ---- weighted recombination, L/:321-326 --------------------------------------
// Block b sums ranks [32 b, 32 b + 32) of the elite list in rank order, one
// thread per coordinate (coalesced row reads); rows that live in another
// context are skipped.  k_recombine_final adds the block partials in block
// order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void k_recombine_partial(const float *__restrict__ rows, int ld,
                                                           int N, int first_candidate, int n_local,
                                                           const int32_t *__restrict__ perm,
                                                           const double *__restrict__ weights,
                                                           int mu, double *__restrict__ block_partials)
{
    const int r0 = blockIdx.x * kRecombineRanksPerBlock;
    const int r1 = min(mu, r0 + kRecombineRanksPerBlock);
    for (int n = threadIdx.x; n < N; n += 256) {
        double acc = 0.0;
        for (int r = r0; r < r1; ++r) {
            const int local = perm[r] - first_candidate;
            if (local >= 0 && local < n_local)
                acc += weights[r] * (double)rows[(size_t)local * ld + n];
        }
        block_partials[(size_t)blockIdx.x * N + n] = acc;
    }
}

__global__ __launch_bounds__(256) void k_recombine_final(const double *__restrict__ block_partials,
                                                         int nblocks, int N,
                                                         double *__restrict__ partial_local)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    for (int b = 0; b < nblocks; ++b) acc += block_partials[(size_t)b * N + n];
    partial_local[n] = acc;
}

// ---- update ---------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__device__ __forceinline__ double wave_dot(const double *a, const double *b, int N, int lane)
{
    double d = 0.0;
    for (int n = lane; n < N; n += 64) d += a[n] * b[n];
    return wave_sum(d);
}

constexpr int kUpdThreads = 1024;

__global__ __launch_bounds__(kUpdThreads) void k_update(const UpdateArgs a)
{
    __shared__ int s_t[LMCMA_HIP_MAX_VECTORS];
    __shared__ int s_imin, s_isz, s_newidx;
    __shared__ long long s_rank_part[kUpdThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nwaves = kUpdThreads / 64;
    const int N = a.N, m = a.m, lambda = a.lambda;
    OptScalars &o = *a.opt;
    const double sigma = o.sigma, cc = o.cc, c1 = o.c1;
    const int itr = o.itr;

    // L/:316-329: new mean from the rank-ordered partial sums, evolution path
    const double pc_gain = sqrt(cc * (2 - cc) * o.mueff);
    for (int n = tid; n < N; n += kUpdThreads) {
        double xnew = 0.0;
        for (int p = 0; p < a.n_partials; ++p) xnew += a.partials[(size_t)p * N + n];
        const double xold = a.xmean[n];
        a.xmean[n] = xnew;
        a.pc[n] = (1 - cc) * a.pc[n] + pc_gain * (xnew - xold) / sigma;
    }

    // L/:331-364: which slot receives the new direction
    if (tid == 0) {
        int imin = 1;
        if (itr < m) {
            a.t[itr] = itr;
        } else if (m > 1) {
            int dmin = a.vec[a.t[1]] - a.vec[a.t[0]];
            for (int j = 1; j < m - 1; ++j) {
                const int dcur = a.vec[a.t[j + 1]] - a.vec[a.t[j]];
                if (dcur < dmin) {
                    dmin = dcur;
                    imin = j + 1;
                }
            }
            if (dmin >= m) imin = 0;  // maxsteps = nvectors (L/:267)
            if (imin != m - 1) {
                const int sav = a.t[imin];
                for (int j = imin; j < m - 1; ++j) a.t[j] = a.t[j + 1];
                a.t[m - 1] = sav;
            }
        } else {
            imin = 0;
        }
        int isz = itr + 1;
        if (isz > m) isz = m;
        for (int i = 0; i < isz; ++i) {
            a.iterator[i] = a.t[i];
            s_t[i] = a.t[i];
        }
        const int newidx = a.t[isz - 1];
        a.vec[newidx] = itr;
        if (imin == 1) imin = 0;  // L/:373-374
        s_imin = imin;
        s_isz = isz;
        s_newidx = newidx;
    }
    __syncthreads();
    const int imin = s_imin, isz = s_isz, newidx = s_newidx;

    // L/:365-366 and the start of the recomputation (:378-379): Av_i <- pc_arr[t[i]]
    for (int n = tid; n < N; n += kUpdThreads) a.pc_arr[(size_t)newidx * N + n] = a.pc[n];
    __syncthreads();
    for (int i = imin + wave; i < isz; i += nwaves) {
        const int slot = s_t[i];
        for (int n = lane; n < N; n += 64)
            a.v_arr[(size_t)slot * N + n] = a.pc_arr[(size_t)slot * N + n];
    }
    __syncthreads();

    // L/:375-390 with invAz (:449-463).  The reference finishes vector i before it
    // starts i+1; vector i only needs the transforms T_j = K I - lj_j v_j v_j^T of the
    // vectors before it, applied in order j = 0..i-1.  Here j is the outer loop and
    // every unfinished vector applies T_j at the same time (one wave per vector), so
    // each vector still sees exactly the reference's sequence of operations.
    const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1), K = o.K;
    for (int i = 0; i < isz; ++i) {
        // vector i has received T_0..T_{i-1} and is final: publish its coefficients
        if (i >= imin && ((i - imin) % nwaves) == wave) {
            const int slot = s_t[i];
            const double *v = a.v_arr + (size_t)slot * N;
            const double nv = wave_dot(v, v, N, lane);
            if (lane == 0) {
                a.nj[slot] = (sq / nv) * (sqrt(1 + ratio * nv) - 1);
                a.lj[slot] = (1 / (sq * nv)) * (1 - (1 / sqrt(1 + ratio * nv)));
            }
        }
        __syncthreads();
        // apply T_i to every later vector that is being recomputed
        const int slot_i = s_t[i];
        const double *vi = a.v_arr + (size_t)slot_i * N;
        const double lj = a.lj[slot_i];
        for (int q = (i + 1 > imin) ? i + 1 : imin; q < isz; ++q) {
            if (((q - imin) % nwaves) != wave) continue;
            double *av = a.v_arr + (size_t)s_t[q] * N;
            const double d = lj * wave_dot(vi, av, N, lane);
            for (int n = lane; n < N; n += 64) av[n] = K * av[n] - d * vi[n];
        }
        __syncthreads();
    }

    // L/:393-419: population success rule.  In the merged stable sort of
    // [current | previous] an element of the current generation at sorted position i
    // gets rank i + #{previous < it}; the reference's "meanprev" is the mean rank of
    // the current generation and "meancur" that of the previous one.
    if (itr > 0) {
        long long part = 0;
        for (int i = tid; i < lambda; i += kUpdThreads) {
            const float c = a.sorted_cost[i];
            int lo = 0, hi = lambda;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.prev_sorted_cost[mid] < c) lo = mid + 1; else hi = mid;
            }
            part += i + lo;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) part += __shfl_xor(part, s, 64);
        if (lane == 0) s_rank_part[wave] = part;
    }
    __syncthreads();
    if (tid == 0) {
        double new_sigma = sigma;
        if (itr > 0) {
            long long cur_sum = 0;
            for (int w = 0; w < nwaves; ++w) cur_sum += s_rank_part[w];
            const long long total = (long long)lambda * (2ll * lambda - 1);  // sum of 0..2*lambda-1
            const double mean_current = (double)cur_sum / lambda;
            const double mean_previous = (double)(total - cur_sum) / lambda;
            const double diffv = (mean_previous - mean_current) / lambda;
            const double z1 = diffv - o.val_target;
            o.s = (1 - o.cs) * o.s + o.cs * z1;
            new_sigma = sigma * exp(o.s / 1.0);
            o.sigma = new_sigma;
        }
        // CMABase::setEvaluationFeedback bookkeeping, L/:191-198, for lambda candidates
        const double best = (double)a.sorted_cost[0];
        if (best < o.best_f || o.counteval == 0) o.best_f = best;
        o.counteval += lambda;
        o.itr = itr + 1;
        o.iterator_sz = isz;
        o.imin_last = imin;
        // scalars of the next sampling pass
        double mk = 1.0;
        for (int q = 0; q < isz; ++q) mk *= o.M;
        a.sc->sigma = (float)new_sigma;
        a.sc->mk = (float)mk;
        a.sc->k = isz;
        a.sc->generation = (uint32_t)(itr + 1);
    }
    __syncthreads();  // the PSR reads of prev_sorted_cost are done

    // L/:420-421 and the float images the sampling kernel reads
    for (int i = tid; i < lambda; i += kUpdThreads) a.prev_sorted_cost[i] = a.sorted_cost[i];
    for (int n = tid; n < a.ld; n += kUpdThreads) a.xmean_f[n] = (n < N) ? (float)a.xmean[n] : 0.f;
    for (int q = wave; q < isz; q += nwaves) {
        const int slot = s_t[q];
        double coef = a.nj[slot];
        for (int e = 0; e < isz - 1 - q; ++e) coef *= o.M;  // M^(k-1-t) nj
        for (int n = lane; n < a.ld; n += 64) {
            const bool in = n < N;
            a.V_f[(size_t)q * a.ld + n] = in ? (float)a.v_arr[(size_t)slot * N + n] : 0.f;
            a.B_f[(size_t)q * a.ld + n] = in ? (float)(coef * a.pc_arr[(size_t)slot * N + n]) : 0.f;
        }
    }
}

}  // namespace

int recombine_num_blocks(int mu)
{
    return (mu + kRecombineRanksPerBlock - 1) / kRecombineRanksPerBlock;
}

hipError_t launch_recombine(hipStream_t s, const float *rows, int ld, int N, int first_candidate,
                            int n_local, const int32_t *perm, const double *weights, int mu,
                            double *block_partials, double *partial_local)
{
    if (mu <= 0 || N <= 0) return hipErrorInvalidValue;
    const int nb = recombine_num_blocks(mu);
    hipLaunchKernelGGL(k_recombine_partial, dim3(nb), dim3(256), 0, s, rows, ld, N, first_candidate,
                       n_local, perm, weights, mu, block_partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_recombine_final, dim3((N + 255) / 256), dim3(256), 0, s, block_partials, nb,
                       N, partial_local);
    return hipGetLastError();
}

hipError_t launch_update(hipStream_t s, const UpdateArgs &a)
{
    if (a.m < 1 || a.m > LMCMA_HIP_MAX_VECTORS || a.N < 1 || a.n_partials < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_update, dim3(1), dim3(kUpdThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace lmcma
