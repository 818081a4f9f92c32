// Elite recombination and the on-device LM-CMA update (fp64).
//
// Replaces LMCMA::update (lmcma_path_planner/src/lmcma.cpp:313-424) and invAz
// (:449-463); the citations L/:a-b below are lines of that file.
#include "device_funcs.hpp"

namespace lmcma {

namespace {

// ---- weighted recombination, L/:321-326 --------------------------------------
// Block b owns ranks [32 b, 32 b + 32) of the elite list.  A thread owns one
// float4 column chunk of one "slot"; slot s walks ranks s, s + nslots, ... of the
// block in order, so all row loads of a block are independent and coalesced.
// The slots are then added in slot order through LDS, and k_update (or
// k_recombine_final) adds the blocks in block order: the summation tree is fixed
// by (mu, ld) alone.  Rows that live in another context are skipped.
__global__ __launch_bounds__(256) void k_recombine_partial(const float *__restrict__ rows, int ld,
                                                           int N, int first_candidate, int n_local,
                                                           const int32_t *__restrict__ perm,
                                                           const double *__restrict__ weights,
                                                           int mu, double *__restrict__ block_partials)
{
    extern __shared__ __attribute__((aligned(16))) double s_part[];  // [nslots][ld]
    const int nq = ld >> 2;               // float4 chunks per row (<= 128)
    const int nslots = 256 / nq;
    const int r0 = blockIdx.x * kRecombineRanksPerBlock;
    const int r1 = min(mu, r0 + kRecombineRanksPerBlock);
    const int slot = threadIdx.x / nq;
    const int q = threadIdx.x % nq;
    if (slot < nslots) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int r = r0 + slot; r < r1; r += nslots) {
            const int local = perm[r] - first_candidate;
            if (local >= 0 && local < n_local) {
                const float4 x = reinterpret_cast<const float4 *>(rows + (size_t)local * ld)[q];
                const double w = weights[r];
                a0 += w * (double)x.x; a1 += w * (double)x.y;
                a2 += w * (double)x.z; a3 += w * (double)x.w;
            }
        }
        double *dst = s_part + (size_t)slot * ld + 4 * q;
        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        double acc = 0.0;
        for (int s = 0; s < nslots; ++s) acc += s_part[(size_t)s * ld + n];
        block_partials[(size_t)blockIdx.x * N + n] = acc;
    }
}

// sum of `count` rows of src[count][N] in row order, 8 threads per coordinate:
// thread j of a coordinate adds rows j, j+8, ...; the 8 partial sums are then
// added in lane order 0..7 (fixed tree).
__device__ __forceinline__ double strided_partial8(const double *__restrict__ src, int count, int N, int n, int j)
{
    double acc = 0.0;
    for (int b0 = j; b0 < count; b0 += 64) {
        // eight independent loads in flight, then added in row order
        double r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 8 * u;
            r[u] = b < count ? src[(size_t)b * N + n] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + 8 * u < count) acc += r[u];
    }
    return acc;
}

// the 8 partial sums of lanes j = 0..7 (consecutive lanes of one wave) added in lane order
__device__ __forceinline__ double lanes8_sum(double acc)
{
    double total = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) total += __shfl(acc, (threadIdx.x & 56) + s, 64);
    return total;
}

__device__ __forceinline__ double strided_sum8(const double *__restrict__ src, int count, int N, int n, int j)
{
    return lanes8_sum(strided_partial8(src, count, N, n, j));
}

__global__ __launch_bounds__(256) void k_recombine_final(const double *__restrict__ block_partials,
                                                         int nblocks, int N,
                                                         double *__restrict__ partial_local)
{
    const int n = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int j = threadIdx.x & 7;
    const int nn = n < N ? n : N - 1;  // idle lanes shadow a valid column (shuffles need all lanes)
    const double total = strided_sum8(block_partials, nblocks, N, nn, j);
    if (n < N && j == 0) partial_local[n] = total;
}

// ---- update ---------------------------------------------------------------------

__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

__device__ __forceinline__ double fast_rsq(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// L/:388-389: lj = (1/(sq nv)) (1 - 1/sqrt(1 + ratio nv)), with the reciprocal and the
// reciprocal square root from the hardware estimate plus two Newton steps (full fp64
// precision at a fraction of the latency of the IEEE divide and sqrt sequences)
__device__ __forceinline__ double lj_of(double nv, double sq, double ratio)
{
    return fast_rcp(sq * nv) * (1 - fast_rsq(1 + ratio * nv));
}

#ifdef LMCMA_DIAG
// diagnostic build only: wall-clock stamps (100 MHz counter) of the phases of k_update
__device__ unsigned long long g_update_stamps[16];
#define STAMP(i)                                                            \
    do {                                                                    \
        if (threadIdx.x == 0) g_update_stamps[i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define STAMP(i)
#endif

// One workgroup; 16 lanes own one direction vector, EMAX coordinates per lane held
// in registers.  V_LDS: finished vectors are published through LDS (m * N * 8
// bytes), otherwise through their home rows in global memory.  Global stores are
// kept out of the loop (a workgroup barrier waits for outstanding stores) and issued
// once at the end.
template <int EMAX, bool V_LDS>
__global__ __launch_bounds__(1024) void k_update(const UpdateArgs a)
{
    // all LDS in one dynamic array, widest elements first, so every carve stays aligned
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    const int T = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;
    const int N = a.N, m = a.m, lambda = a.lambda;
    OptScalars &o = *a.opt;
    const double sigma = o.sigma, cc = o.cc, c1 = o.c1;
    const int itr = o.itr;

    double *s_v = s_dyn;                                        // [m][N] when V_LDS
    double *s_pc = s_dyn + (V_LDS ? (size_t)m * N : 0);         // [N] new evolution path
    double *s_xm = s_pc + N;                                    // [N] new mean
    double *s_lj = s_xm + N;                                    // [64], slot-indexed
    double *s_nj = s_lj + LMCMA_HIP_MAX_VECTORS;                // [64], slot-indexed
    double *s_nv = s_nj + LMCMA_HIP_MAX_VECTORS;                // [64], |v_i|^2 by position i
    double *s_ljp = s_nv + LMCMA_HIP_MAX_VECTORS;               // [64], lj of the untouched vectors by position
    int *s_t = reinterpret_cast<int *>(s_ljp + LMCMA_HIP_MAX_VECTORS);  // [64]
    int *s_vec = s_t + LMCMA_HIP_MAX_VECTORS;                   // [64]
    int *s_misc = s_vec + LMCMA_HIP_MAX_VECTORS;                // imin, isz, newidx
    int &s_imin = s_misc[0], &s_isz = s_misc[1], &s_newidx = s_misc[2];

    STAMP(0);
#ifdef LMCMA_DIAG
    if (threadIdx.x == 0) g_update_stamps[10] = __builtin_amdgcn_s_memtime();
#endif
    if (tid < m) {
        s_t[tid] = a.t[tid];
        s_vec[tid] = a.vec[tid];
        s_lj[tid] = a.lj[tid];
        s_nj[tid] = a.nj[tid];
    }
    // inputs of the step-size rule, fetched early (thread 0 uses them at the end)
    unsigned long long psr_in = 0ull;
    float best_in = 0.f;
    if (tid == 0) {
        psr_in = *a.psr_accum;
        best_in = a.sorted_cost[0];
    }
    __syncthreads();
    STAMP(1);

    // L/:331-364: which slot receives the new direction (needs only the slot tables,
    // so it runs first and the vector loads below overlap the partial-sum loads)
    if (tid == 0) {
        int imin = 1;
        if (itr < m) {
            s_t[itr] = itr;
        } else if (m > 1) {
            int dmin = s_vec[s_t[1]] - s_vec[s_t[0]];
            for (int j = 1; j < m - 1; ++j) {
                const int dcur = s_vec[s_t[j + 1]] - s_vec[s_t[j]];
                if (dcur < dmin) {
                    dmin = dcur;
                    imin = j + 1;
                }
            }
            if (dmin >= m) imin = 0;  // maxsteps = nvectors (L/:267)
            if (imin != m - 1) {
                const int sav = s_t[imin];
                for (int j = imin; j < m - 1; ++j) s_t[j] = s_t[j + 1];
                s_t[m - 1] = sav;
            }
        } else {
            imin = 0;
        }
        int isz = itr + 1;
        if (isz > m) isz = m;
        const int newidx = s_t[isz - 1];
        s_vec[newidx] = itr;
        if (imin == 1) imin = 0;  // L/:373-374
        s_imin = imin;
        s_isz = isz;
        s_newidx = newidx;
    }
    __syncthreads();
    STAMP(2);
    const int imin = s_imin, isz = s_isz, newidx = s_newidx;

    // One 16-lane group per stored direction; the loads of its vector and of its
    // evolution path are issued here, ahead of the partial-sum reduction.
    const bool own = grp < isz;
    const int slot = own ? s_t[grp] : 0;
    const bool rebuild = own && grp >= imin;
    const bool newest = own && slot == newidx;  // L/:365-366: takes the new evolution path
    double av[EMAX], pcq[EMAX];  // the vector and its evolution path pc_arr[slot]
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int n = e * 16 + lane;
        const bool in = own && n < N;
        pcq[e] = (in && !newest) ? a.pc_arr[(size_t)slot * N + n] : 0.0;
        av[e] = (in && !rebuild) ? a.v_arr[(size_t)slot * N + n] : 0.0;
    }

    // L/:316-329: new mean from the rank-ordered partial sums, evolution path.
    // 8 threads per coordinate add the partials with the summation tree of strided_sum8;
    // the loads of all passes are issued before the first cross-lane step.
    const double pc_gain = sqrt(cc * (2 - cc) * o.mueff);
    {
        constexpr int kPasses = 4;  // 1024 threads / 8 = 128 coordinates per pass, N <= 512
        double acc[kPasses];
        const int j = tid & 7;
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {
            const int n = ps * (T / 8) + (tid >> 3);
            acc[ps] = (ps * (T / 8) < N) ? strided_partial8(a.partials, a.n_partials, N, n < N ? n : N - 1, j) : 0.0;
        }
#ifdef LMCMA_DIAG
        if (acc[0] == 1.2345e300) g_update_stamps[15] = 1;  // forces the loads to complete here
        STAMP(8);
#endif
#pragma unroll
        for (int ps = 0; ps < kPasses; ++ps) {
            if (ps * (T / 8) < N) {  // uniform
                const int n = ps * (T / 8) + (tid >> 3);
                const double xnew = lanes8_sum(acc[ps]);
#ifdef LMCMA_DIAG
                if (xnew == 1.2345e300) g_update_stamps[15] = 2;
                if (ps == 0) STAMP(9);
#endif
                if (n < N && j == 0) {
                    const double xold = a.xmean[n];
                    const double pcn = (1 - cc) * a.pc[n] + pc_gain * (xnew - xold) / sigma;
                    s_pc[n] = pcn;
                    s_xm[n] = xnew;
                }
            }
        }
    }
    __syncthreads();
    STAMP(3);

    // L/:375-390 with invAz (:449-463).  Vector i is v_i = T_{i-1} ... T_0 pc_i with
    // T_j = K I - lj_j v_j v_j^T.  The reference finishes vector i before it starts
    // i+1; here j is the outer loop and every unfinished vector applies T_j at the same
    // time (one 16-lane group per vector, the vector in registers), so each vector
    // still sees exactly the reference's sequence of operations.  Vectors before imin
    // keep last generation's value (L/:375 starts at imin).
    //
    // The squared norm of every vector is carried along instead of being summed again
    // after the last transform: |K a - d v|^2 = K^2 |a|^2 - 2 K d (v.a) + d^2 |v|^2, all of
    // whose terms are already in registers; so a step needs one cross-lane sum only.
    double aa = 0.0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int n = e * 16 + lane;
        if (newest && n < N) pcq[e] = s_pc[n];
        if (rebuild) av[e] = pcq[e];
        aa += av[e] * av[e];
        if (V_LDS && own && !rebuild && n < N) s_v[(size_t)grp * N + n] = av[e];
    }
    aa = row16_sum(aa);
    if (own && !rebuild && lane == 0) {
        s_nv[grp] = aa;
        s_ljp[grp] = s_lj[slot];
    }
    __syncthreads();
    STAMP(4);
    const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1), K = o.K;
    for (int i = 0; i < isz; ++i) {
        const bool fresh = i >= imin;  // same for every thread
        if (fresh) {
            // vector i has received T_0..T_{i-1}: it is final, its owner publishes it and its
            // squared norm; vectors before imin were published above, once
            if (grp == i) {
                double *dst = V_LDS ? s_v + (size_t)i * N : a.v_arr + (size_t)slot * N;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    const int n = e * 16 + lane;
                    if (n < N) dst[n] = av[e];
                }
                if (lane == 0) s_nv[i] = aa;
            }
            // The next publication writes row i+1 and another norm word, so one barrier per
            // step is enough.
            __syncthreads();
        }
        // T_i on every later vector that is being rebuilt
        if (rebuild && grp > i) {
            const double *vi = V_LDS ? s_v + (size_t)i * N : a.v_arr + (size_t)s_t[i] * N;
            double vv[EMAX];
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int e = 0; e < EMAX; e += 2) {
                const int n0 = e * 16 + lane, n1 = (e + 1) * 16 + lane;
                vv[e] = n0 < N ? vi[n0] : 0.0;
                vv[e + 1] = n1 < N ? vi[n1] : 0.0;
                p0 += vv[e] * av[e];
                p1 += vv[e + 1] * av[e + 1];
            }
            const double nvi = s_nv[i];
            // every group derives lj_i from the published norm itself (arithmetic that hides
            // behind the LDS reads above) instead of waiting for the owner to do it
            const double lji = fresh ? lj_of(nvi, sq, ratio) : s_ljp[i];
            const double p = row16_sum(p0 + p1);
            const double d = lji * p;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) av[e] = K * av[e] - d * vv[e];
            aa = K * K * aa - 2.0 * K * d * p + d * d * nvi;
        }
    }
    // L/:386-389 for the rebuilt vectors
    if (rebuild && lane == 0) {
        s_nj[slot] = (sq / aa) * (sqrt(1 + ratio * aa) - 1);
        s_lj[slot] = lj_of(aa, sq, ratio);
    }
    __syncthreads();
    STAMP(5);

    // L/:393-419: population success rule.  In the merged stable sort of
    // [current | previous] an element of the current generation at sorted position i
    // gets rank i + #{previous < it}; the ranking kernels have summed the second term
    // into psr_accum.  The reference's "meanprev" is the mean rank of the current
    // generation and "meancur" that of the previous one.
    if (tid == 0) {
        double new_sigma = sigma;
        if (itr > 0) {
            const long long cur_sum = (long long)lambda * (lambda - 1) / 2 + (long long)psr_in;
            const long long total = (long long)lambda * (2ll * lambda - 1);  // sum of 0..2*lambda-1
            const double mean_current = (double)cur_sum / lambda;
            const double mean_previous = (double)(total - cur_sum) / lambda;
            const double diffv = (mean_previous - mean_current) / lambda;
            const double z1 = diffv - o.val_target;
            o.s = (1 - o.cs) * o.s + o.cs * z1;
            new_sigma = sigma * exp(o.s / 1.0);
            o.sigma = new_sigma;
        }
        *a.psr_accum = 0ull;  // ready for the next generation's ranking
        // CMABase::setEvaluationFeedback bookkeeping, L/:191-198, for lambda candidates
        const double best = (double)best_in;
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
    STAMP(6);

    // state back to memory, and the float images the sampling kernel reads:
    // V_f[q] = v_q, B_f[q] = M^(k-1-q) nj_q pc_q
    for (int n = tid; n < a.ld; n += T) {
        if (n < N) {
            a.xmean[n] = s_xm[n];
            a.pc[n] = s_pc[n];
        }
        a.xmean_f[n] = (n < N) ? (float)s_xm[n] : 0.f;
    }
    if (tid < m) {
        a.t[tid] = s_t[tid];
        a.vec[tid] = s_vec[tid];
        a.nj[tid] = s_nj[tid];
        a.lj[tid] = s_lj[tid];
        if (tid < isz) a.iterator[tid] = s_t[tid];
    }
    if (own) {
        double coef = s_nj[slot];
        const double Mfac = o.M;
        for (int e = 0; e < isz - 1 - grp; ++e) coef *= Mfac;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            if (n < N) {
                if (slot == newidx) a.pc_arr[(size_t)slot * N + n] = pcq[e];
                if (rebuild && V_LDS) a.v_arr[(size_t)slot * N + n] = av[e];
            }
            if (n < a.ld) {
                a.V_f[(size_t)grp * a.ld + n] = (n < N) ? (float)av[e] : 0.f;
                a.B_f[(size_t)grp * a.ld + n] = (n < N) ? (float)(coef * pcq[e]) : 0.f;
            }
        }
    }
    STAMP(7);
#ifdef LMCMA_DIAG
    if (threadIdx.x == 0) g_update_stamps[11] = __builtin_amdgcn_s_memtime();
#endif
}

}  // namespace

#ifdef LMCMA_DIAG
extern "C" hipError_t lmcma_diag_update_stamps(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_update_stamps), 16 * sizeof(unsigned long long));
}
#endif

int recombine_num_blocks(int mu)
{
    return (mu + kRecombineRanksPerBlock - 1) / kRecombineRanksPerBlock;
}

hipError_t launch_recombine(hipStream_t s, const float *rows, int ld, int N, int first_candidate,
                            int n_local, const int32_t *perm, const double *weights, int mu,
                            double *block_partials, double *partial_local)
{
    if (mu <= 0 || N <= 0 || (ld & 3) || ld > 512) return hipErrorInvalidValue;
    const int nb = recombine_num_blocks(mu);
    const int nslots = 256 / (ld >> 2);
    const size_t lds = (size_t)nslots * ld * sizeof(double);
    hipLaunchKernelGGL(k_recombine_partial, dim3(nb), dim3(256), lds, s, rows, ld, N, first_candidate,
                       n_local, perm, weights, mu, block_partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !partial_local) return e;
    hipLaunchKernelGGL(k_recombine_final, dim3((N + 31) / 32), dim3(256), 0, s, block_partials, nb, N,
                       partial_local);
    return hipGetLastError();
}

hipError_t launch_recombine_final(hipStream_t s, const double *block_partials, int rows, int N,
                                  double *partial_local)
{
    if (rows < 1 || N < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_recombine_final, dim3((N + 31) / 32), dim3(256), 0, s, block_partials, rows, N,
                       partial_local);
    return hipGetLastError();
}

template <int EMAX, bool V_LDS>
static hipError_t launch_update_t(hipStream_t s, const UpdateArgs &a, size_t lds, int threads)
{
    static size_t configured = 48 * 1024;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_update<EMAX, V_LDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL((k_update<EMAX, V_LDS>), dim3(1), dim3(threads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_update(hipStream_t s, const UpdateArgs &a)
{
    if (a.m < 1 || a.m > LMCMA_HIP_MAX_VECTORS || a.N < 1 || a.N > 512 || a.n_partials < 1)
        return hipErrorInvalidValue;
    const size_t fixed = (4 * LMCMA_HIP_MAX_VECTORS + 2 * (size_t)a.N) * sizeof(double) +
                         (2 * LMCMA_HIP_MAX_VECTORS + 4) * sizeof(int);
    const size_t v_bytes = (size_t)a.m * a.N * sizeof(double);
    const bool v_lds = v_bytes + fixed <= 150 * 1024;
    const size_t lds = fixed + (v_lds ? v_bytes : 0);
    const int threads = 1024;  // 16 lanes per direction vector (m <= 64), 8 lanes per coordinate in the mean
    const int emax = (a.N + 15) / 16;
#define LMCMA_UPD(E)                                                             \
    return v_lds ? launch_update_t<E, true>(s, a, lds, threads) : launch_update_t<E, false>(s, a, lds, threads)
    if (emax <= 8) { LMCMA_UPD(8); }
    if (emax <= 16) { LMCMA_UPD(16); }
    LMCMA_UPD(32);
#undef LMCMA_UPD
}

}  // namespace lmcma
