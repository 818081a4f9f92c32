// The LM-CMA state update as a device function of one workgroup (fp64), and the
// fixed-order row sums it shares with the ranking kernels.
//
// Replaces LMCMA::update (lmcma_path_planner/src/lmcma.cpp:313-424) and invAz
// (:449-463); the citations L/:a-b below are lines of that file.
#pragma once

#include "device_funcs.hpp"

namespace lmcma {

// the argument block of query q (Batch): every context buffer moves by q * stride bytes
__device__ __forceinline__ UpdateArgs shifted(const UpdateArgs &in, int q)
{
    UpdateArgs a = in;
    const size_t off = (size_t)q * in.batch.stride;
    if (off == 0) return a;
    shift_ptr(a.opt, off); shift_ptr(a.sc, off); shift_ptr(a.partials, off); shift_ptr(a.sorted_cost, off);
    shift_ptr(a.psr_accum, off); shift_ptr(a.psr_parts, off); shift_ptr(a.xmean, off); shift_ptr(a.pc, off);
    shift_ptr(a.pc_arr, off); shift_ptr(a.v_arr, off); shift_ptr(a.nj, off); shift_ptr(a.lj, off);
    shift_ptr(a.t, off); shift_ptr(a.vec, off); shift_ptr(a.iterator, off);
    shift_ptr(a.xmean_f, off); shift_ptr(a.V_f, off); shift_ptr(a.B_f, off);
    return a;
}

// ---- rows of doubles added in a fixed order by one workgroup ------------------------
// Thread (r, n) adds rows r, r + rg, r + 2 rg, ... of column n (rg row groups of cw
// threads each); the rg partial sums of a column are then added in group order.  The
// tree depends on (blockDim, N, count) only.

struct RowSumPlan {
    int cw;  // threads per row group, at most the workgroup
    int rg;  // row groups
};

__device__ __forceinline__ RowSumPlan row_sum_plan(int N, int T)
{
    int cw = (N + 63) & ~63;
    if (cw > T) cw = T;
    return {cw, T / cw};
}

// The same with two adjacent columns per thread (16-byte loads), for even N.
__device__ __forceinline__ RowSumPlan row_sum_plan2(int N, int T)
{
    int cw = ((N >> 1) + 63) & ~63;
    if (cw > T) cw = T;
    return {cw, T / cw};
}

// first half: partial sums to s_red[r * N + n]; the caller puts a barrier after it.
// The rows were written by the previous kernel, so every first touch misses this XCD's
// L2, and one workgroup has to keep many wide loads in flight to read them at any speed:
// 16 bytes per lane (even N; a single workgroup reads such data about twice as fast with
// 16-byte as with 8-byte loads) and up to 16 loads per thread at once.
constexpr int kRowBatch = 16;

// The first batch of loads of row_sums_to_lds (even N), issued early: a caller with other
// work to do between the request and the use calls row_sums_begin, does that work, then
// row_sums_to_lds with the same arguments and the batch.
struct RowSumBatch {
    double2 v[kRowBatch];
};

__device__ __forceinline__ void row_sums_begin(const double *src, int count, int N, int stride, int T, RowSumBatch &first)
{
#pragma unroll
    for (int u = 0; u < kRowBatch; ++u) first.v[u] = make_double2(0.0, 0.0);
    if (((N | stride) & 1) != 0) return;
    const RowSumPlan pl = row_sum_plan2(N, T);
    const int r = __builtin_amdgcn_readfirstlane((int)threadIdx.x / pl.cw);
    const int n = 2 * ((int)threadIdx.x % pl.cw);
    if (r >= pl.rg || n >= N) return;
#pragma unroll
    for (int u = 0; u < kRowBatch; ++u) {
        const int b = r + u * pl.rg;  // wave-uniform
        if (b < count) first.v[u] = *reinterpret_cast<const double2 *>(src + (size_t)b * stride + n);
    }
}

template <bool HAVE_FIRST>
__device__ __forceinline__ void row_sums_to_lds_t(const double *src, int count, int N, int stride, int T,
                                                  double *s_red, const RowSumBatch &first)
{
    constexpr int kBatch = kRowBatch;
    if (((N | stride) & 1) == 0) {
        const RowSumPlan pl = row_sum_plan2(N, T);
        // cw is a multiple of 64 (or the whole workgroup), so r is the same in every lane of a wave
        const int r = __builtin_amdgcn_readfirstlane((int)threadIdx.x / pl.cw);
        const int n0 = 2 * ((int)threadIdx.x % pl.cw);
        if (r >= pl.rg) return;
        for (int n = n0; n < N; n += 2 * pl.cw) {
            double a0 = 0.0, a1 = 0.0;
            for (int b0 = r; b0 < count; b0 += kBatch * pl.rg) {
                double2 v[kBatch];
                if (HAVE_FIRST && n == n0 && b0 == r) {
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) v[u] = first.v[u];
                } else {
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const int b = b0 + u * pl.rg;  // wave-uniform
                        v[u] = make_double2(0.0, 0.0);
                        if (b < count) v[u] = *reinterpret_cast<const double2 *>(src + (size_t)b * stride + n);
                    }
                    asm volatile("" ::: "memory");  // all loads of the batch are issued before the first add
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    if (b0 + u * pl.rg < count) {
                        a0 += v[u].x;
                        a1 += v[u].y;
                    }
            }
            s_red[r * N + n] = a0;
            s_red[r * N + n + 1] = a1;
        }
        return;
    }
    const RowSumPlan pl = row_sum_plan(N, T);
    const int r = __builtin_amdgcn_readfirstlane((int)threadIdx.x / pl.cw);
    const int n0 = (int)threadIdx.x % pl.cw;
    if (r >= pl.rg) return;
    for (int n = n0; n < N; n += pl.cw) {
        double acc = 0.0;
        for (int b0 = r; b0 < count; b0 += kBatch * pl.rg) {
            double v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int b = b0 + u * pl.rg;  // wave-uniform
                v[u] = 0.0;
                if (b < count) v[u] = src[(size_t)b * stride + n];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
                if (b0 + u * pl.rg < count) acc += v[u];
        }
        s_red[r * N + n] = acc;
    }
}

__device__ __forceinline__ void row_sums_to_lds(const double *src, int count, int N, int stride, int T, double *s_red)
{
    RowSumBatch none;
    row_sums_to_lds_t<false>(src, count, N, stride, T, s_red, none);
}

// row groups row_sums_to_lds uses, and the doubles of LDS it needs
__host__ __device__ inline int row_sum_groups(int N, int T)
{
    int cw = (((N & 1) == 0 ? N >> 1 : N) + 63) & ~63;
    if (cw > T) cw = T;
    return T / cw;
}

__host__ __device__ inline size_t row_sum_lds_doubles(int N, int T) { return (size_t)row_sum_groups(N, T) * N; }

// second half: the rg partial sums of column n, added in group order
__device__ __forceinline__ double row_sums_total(const double *s_red, int N, int n, int rg)
{
    double total = 0.0;
    for (int r = 0; r < rg; ++r) total += s_red[r * N + n];
    return total;
}

// ---- scalar helpers -----------------------------------------------------------------

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

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int o = __shfl_xor(v, s, 64);
        v = o < v ? o : v;
    }
    return v;
}

#ifdef LMCMA_DIAG
// diagnostic build only: wall-clock stamps (100 MHz counter) of kernel phases
static __device__ unsigned long long g_update_stamps[16];
// != 0: workgroup 1 of a sampling launch that carries the update does not report to workgroup 0,
// whose wait then gives up (the path a test has to be able to force)
static __device__ int g_diag_skip_report;
#define LMCMA_STAMP(i)                                                                    \
    do {                                                                                  \
        if (threadIdx.x == 0) g_update_stamps[i] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)
#else
#define LMCMA_STAMP(i)
#endif

// Coordinates per lane a 16-lane group keeps in registers for a vector of N parameters
// (N <= 512): the register tile of the update kernels, and 1/16 of their LDS row stride
__host__ __device__ constexpr int update_emax(int N)
{
    return N <= 128 ? 8 : N <= 160 ? 10 : N <= 192 ? 12 : N <= 256 ? 16 : N <= 320 ? 20 : N <= 384 ? 24 : 32;
}

// LDS doubles the update needs for (N, m) when TR threads read the partial rows
__host__ __device__ inline size_t update_lds_doubles(int N, int m, int TR, bool v_lds)
{
    const size_t np = v_lds ? 16 * (size_t)update_emax(N) : 512;  // without LDS vectors: the widest tile only
    return (v_lds ? (size_t)m * np : 0) + np + row_sum_lds_doubles(N, TR) + 2 * LMCMA_HIP_MAX_VECTORS;
}

// T >= 16 m threads (T a multiple of 64) of one workgroup do the vector work; when the
// workgroup has more waves, the first of them evaluates the step-size rule beside the
// workers (its divisions and exponential would otherwise hold up a wave the others wait
// for).  The first TR threads of the workgroup (TR >= T) read the partial rows of the new
// mean: a single workgroup reads freshly written rows faster the more loads it has in
// flight, so the extra waves exist mainly for that.  The 16 lanes
// of group g own SLOT g of the direction store (EMAX >= N / 16 coordinates per lane in
// registers), so every load of the kernel can be issued at its start: nothing waits for
// the slot tables.  V_LDS: finished vectors are published through LDS (m * N doubles,
// by position in the time-ordered list), otherwise through their home rows in global
// memory.  Global stores are kept out of the loop (a workgroup barrier waits for
// outstanding stores) and issued at the end.
template <int EMAX, bool V_LDS>
__device__ __forceinline__ void update_body(const UpdateArgs &a, double *s_dyn, int T, int TR)
{
    constexpr bool KEEP_PC = EMAX <= 16;  // evolution paths stay in registers when they fit (20 per lane spill)
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    const int N = a.N, m = a.m, lambda = a.lambda;
    OptScalars &o = *a.opt;
    // The stop rule (L/:426-429) is looked at only after every load of the kernel has been
    // requested (below): a return here would make those requests wait for this scalar's round trip.
    const int done = o.done;
    const double sigma = o.sigma, cc = o.cc, c1 = o.c1, K = o.K, Mfac = o.M;
    const int itr = o.itr;
    const bool worker = tid < T;
    const int stid = (int)blockDim.x > T ? T : 0;  // the thread that evaluates the step-size rule
    const bool swave = (tid >> 6) == (stid >> 6);
    // read now, used by one thread in the step-size rule: a load issued there would have to
    // wait for every load requested before it
    const double o_mueff = o.mueff, o_s = o.s, o_cs = o.cs, o_val_target = o.val_target, o_best_f = o.best_f;
    const long long o_counteval = o.counteval;

    constexpr int NP = EMAX * 16;                          // row stride in LDS: reads and writes need no bounds
    double *s_v = s_dyn;                                  // [m][NP] when V_LDS, by position; zero beyond N
    double *s_pc = s_dyn + (V_LDS ? (size_t)m * NP : 0);  // [NP] new evolution path, zero beyond N
    double *s_red = s_pc + NP;                            // [rg][N] partial column sums
    double *s_nv = s_red + row_sum_lds_doubles(N, TR);              // [64], |v_i|^2 by position i
    double *s_ljp = s_nv + LMCMA_HIP_MAX_VECTORS;         // [64], lj of the untouched vectors by position

    LMCMA_STAMP(0);
    // ---- every load of the kernel
    // slot tables, one entry per lane, the same copy in every wave (m <= 64): position j
    // of the time-ordered list is slot t[j]; slot s was written in generation vec[s]
    int t_reg = 0, vec_reg = 0;
    if (wl < m) {
        t_reg = a.t[wl];
        vec_reg = a.vec[wl];
    }
    const bool has_slot = grp < m;
    const int slot = has_slot ? grp : 0;
    const double lj_old = a.lj[slot], nj_old = a.nj[slot];
    // old mean and evolution path of this thread's columns (N <= 512, T >= 256: at most two)
    double xold[2] = {0.0, 0.0}, pcold[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = tid + q * T;
        if (worker && n < N) {
            xold[q] = a.xmean[n];
            pcold[q] = a.pc[n];
        }
    }
    // success-rule count (the sum over this generation's candidates of the number of last
    // generation's costs below it): one wave reads the per-workgroup shares, up to 1024 of them
    uint32_t pp[16];
    unsigned long long psr_lane = 0ull;
    float best_in = 0.f;
    if (swave) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // no branch around the load (entries past the end re-read word 0 and are dropped
            // below): sixteen independent requests, none waits for another
            const int b = wl + 64 * u;
            pp[u] = a.psr_parts[b < a.n_psr_parts ? b : 0];
        }
        if (a.scalars_in_rows) {
            // a sharded population: every context's row carries its count and the lowest cost
            if (wl < a.n_partials) psr_lane = (unsigned long long)a.partials[(size_t)wl * a.partial_stride + N];
            if (tid == stid) best_in = (float)a.partials[N + 1];
        } else if (tid == stid) {
            psr_lane = *a.psr_accum;
            best_in = a.sorted_cost[0];  // this generation's lowest cost
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) pp[u] = 0u;
    }
    double vold[EMAX], pcq[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) pcq[e] = vold[e] = 0.0;
    if (worker) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            const size_t at = (size_t)slot * N + (n < N ? n : N - 1);
            pcq[e] = a.pc_arr[at];
            vold[e] = a.v_arr[at];
        }
    }
    // L/:316-329: the rank-ordered partial sums of the new mean; the slot bookkeeping and the
    // step-size rule below run while these loads are in flight
    RowSumBatch first_rows;
    row_sums_begin(a.partials, a.n_partials, N, a.partial_stride, TR, first_rows);
    asm volatile("" ::: "memory");  // no load above is moved below this point
    if (done) return;  // the stop rule has fired: the state stays as it is
    LMCMA_STAMP(1);

    // ---- L/:331-364: which slot receives the new direction.  Every wave works it out for
    // itself in registers.
    int imin = 1;
    if (itr < m) {
        if (wl == itr) t_reg = itr;
    } else if (m > 1) {
        const int vt = __shfl(vec_reg, t_reg, 64);  // vec[t[j]] in lane j
        const int vnext = __shfl_down(vt, 1, 64);
        const int gap = (wl < m - 1) ? vnext - vt : 0x7fffffff;
        const int dmin = wave_min_i32(gap);
        // first position with the smallest gap (the reference replaces on strict "<" only)
        imin = __ffsll((unsigned long long)__ballot(gap == dmin));
        if (dmin >= m) imin = 0;  // maxsteps = nvectors (L/:267)
        if (imin != m - 1) {
            const int sav = __shfl(t_reg, imin, 64);
            const int nxt = __shfl_down(t_reg, 1, 64);
            if (wl >= imin && wl < m - 1) t_reg = nxt;
            if (wl == m - 1) t_reg = sav;
        }
    } else {
        imin = 0;
    }
    const int isz = itr + 1 < m ? itr + 1 : m;
    const int newidx = __shfl(t_reg, isz - 1, 64);
    if (wl == newidx) vec_reg = itr;
    if (imin == 1) imin = 0;  // L/:373-374

    // position of this group's slot in the time-ordered list (-1: the slot is not in use)
    int pos = -1;
    {
        const int g0 = (tid >> 6) << 2;  // first group of this wave
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long hit = __ballot(wl < isz && t_reg == g0 + q);
            if ((grp & 3) == q && hit != 0ull) pos = __ffsll(hit) - 1;
        }
    }
    const bool own = has_slot && pos >= 0;
    const bool rebuild = own && pos >= imin;
    const bool newest = own && slot == newidx;  // L/:365-366: takes the new evolution path

    // success-rule count: wave 0 adds its lanes' shares
#pragma unroll
    for (int u = 0; u < 16; ++u) psr_lane += (!a.scalars_in_rows && wl + 64 * u < a.n_psr_parts) ? pp[u] : 0u;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) psr_lane += __shfl_xor(psr_lane, sft, 64);

    const double pc_gain = sqrt(cc * (2 - cc) * o_mueff);
    const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1);

    // L/:393-419: population success rule, by one thread.  In the merged stable sort of
    // [current | previous] an element of the current generation at sorted position i gets
    // rank i + #{previous < it}; psr_lane is the sum of the second term.  The reference's
    // "meanprev" is the mean rank of the current generation and "meancur" that of the
    // previous one.
    double new_sigma = sigma, new_s = 0.0, new_best = 0.0, mk = 1.0;
    long long new_counteval = 0;
    if (tid == stid) {
        new_s = o_s;
        if (itr > 0) {
            const long long cur_sum = (long long)lambda * (lambda - 1) / 2 + (long long)psr_lane;
            const long long total = (long long)lambda * (2ll * lambda - 1);  // sum of 0..2*lambda-1
            const double mean_current = (double)cur_sum / lambda;
            const double mean_previous = (double)(total - cur_sum) / lambda;
            const double diffv = (mean_previous - mean_current) / lambda;
            const double z1 = diffv - o_val_target;
            new_s = (1 - o_cs) * new_s + o_cs * z1;
            new_sigma = sigma * exp(new_s / 1.0);
        }
        // CMABase::setEvaluationFeedback bookkeeping, L/:191-198, for lambda candidates
        const double best = (double)best_in;
        new_counteval = o_counteval;
        new_best = (best < o_best_f || new_counteval == 0) ? best : o_best_f;
        new_counteval += lambda;
        for (int q = 0; q < isz; ++q) mk *= Mfac;  // M^k of the next sampling pass
    }

    // vectors that are not rebuilt are final already: publish them and their norms
    double av[EMAX];
    double aa = 0.0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int n = e * 16 + lane;
        const bool in = own && n < N;
        if (!in) pcq[e] = 0.0;
        av[e] = in ? (rebuild ? pcq[e] : vold[e]) : 0.0;  // L/:376-377: a rebuilt vector starts from its path
    }
    if (!rebuild) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            aa += av[e] * av[e];
            if (V_LDS && own) s_v[(size_t)pos * NP + n] = av[e];
        }
        aa = row16_sum(aa);
        if (own && lane == 0) {
            s_nv[pos] = aa;
            s_ljp[pos] = lj_old;
        }
    }
    row_sums_to_lds_t<true>(a.partials, a.n_partials, N, a.partial_stride, TR, s_red, first_rows);
    __syncthreads();
    LMCMA_STAMP(2);
    // L/:316-329: new mean, evolution path
    double xnew[2] = {0.0, 0.0}, pcn[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = tid + q * T;
        if (worker && n < N) {
            xnew[q] = row_sums_total(s_red, N, n, row_sum_groups(N, TR));
            pcn[q] = (1 - cc) * pcold[q] + pc_gain * (xnew[q] - xold[q]) / sigma;
            s_pc[n] = pcn[q];
        }
    }
    for (int n = N + tid; n < NP; n += (int)blockDim.x) s_pc[n] = 0.0;
    __syncthreads();
    LMCMA_STAMP(3);

    // L/:375-390 with invAz (:449-463).  Vector i is v_i = T_{i-1} ... T_0 pc_i with
    // T_j = K I - lj_j v_j v_j^T.  The reference finishes vector i before it starts
    // i+1; here j is the outer loop and every unfinished vector applies T_j at the same
    // time (one 16-lane group per vector, the vector in registers), so each vector
    // still sees exactly the reference's sequence of operations.  Vectors before imin
    // keep last generation's value (L/:375 starts at imin).
    //
    // The squared norm of a vector is summed when the vector is final, by its owner, just
    // before it is published.  (Carrying it through the transforms as |K a - d v|^2 =
    // K^2 |a|^2 - 2 K d (v.a) + d^2 |v|^2 saves that sum but cancels: with 32 correlated
    // paths the vectors drifted to 2e-9 of the fp64 recurrence, measured.)
    if (newest) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            av[e] = s_pc[e * 16 + lane];
            pcq[e] = av[e];
        }
    }
    if (!KEEP_PC) {
        // the paths are read again at the end; keep the registers for the loop
#pragma unroll
        for (int e = 0; e < EMAX; ++e) pcq[e] = 0.0;
    }
    for (int i = 0; i < isz; ++i) {
        const bool fresh = i >= imin;  // same for every thread
        const int slot_i = __shfl(t_reg, i, 64);
        if (fresh) {
            // vector i has received T_0..T_{i-1}: it is final, its owner publishes it and its
            // squared norm; vectors before imin were published above, once
            if (pos == i) {
                double *dst = V_LDS ? s_v + (size_t)i * NP : a.v_arr + (size_t)slot * N;
                double q0 = 0.0, q1 = 0.0;
#pragma unroll
                for (int e = 0; e < EMAX; e += 2) {
                    const int n0 = e * 16 + lane, n1 = (e + 1) * 16 + lane;
                    if (V_LDS || n0 < N) dst[n0] = av[e];
                    if (V_LDS || n1 < N) dst[n1] = av[e + 1];
                    q0 += av[e] * av[e];
                    q1 += av[e + 1] * av[e + 1];
                }
                aa = row16_sum(q0 + q1);
                if (lane == 0) s_nv[i] = aa;
            }
            // The next publication writes row i+1 and another norm word, so one barrier per
            // step is enough.
            __syncthreads();
        }
        // T_i on every later vector that is being rebuilt
        if (rebuild && pos > i) {
            const double *vi = V_LDS ? s_v + (size_t)i * NP : a.v_arr + (size_t)slot_i * N;
            double vv[EMAX];
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int e = 0; e < EMAX; e += 2) {
                const int n0 = e * 16 + lane, n1 = (e + 1) * 16 + lane;
                vv[e] = (V_LDS || n0 < N) ? vi[n0] : 0.0;
                vv[e + 1] = (V_LDS || n1 < N) ? vi[n1] : 0.0;
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
        }
    }
    LMCMA_STAMP(4);

    // L/:386-389 for the rebuilt vectors (all 16 lanes of a group hold the same norm)
    double nj_new = nj_old;
    if (rebuild) nj_new = (sq / aa) * (sqrt(1 + ratio * aa) - 1);

    // state back to memory, and the float images the sampling kernel reads, by position q:
    // V_f[q] = v_q, B_f[q] = M^(k-1-q) nj_q pc_q
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = tid + q * T;
        if (worker && n < N) {
            a.xmean[n] = xnew[q];
            a.pc[n] = pcn[q];
        }
        if (worker && n < a.ld) a.xmean_f[n] = (n < N) ? (float)xnew[q] : 0.f;
    }
    if (tid < m) {
        a.t[tid] = t_reg;
        a.vec[tid] = vec_reg;
        if (tid < isz) a.iterator[tid] = t_reg;
    }
    if (own) {
        if (rebuild && lane == 0) {
            a.nj[slot] = nj_new;
            a.lj[slot] = lj_of(aa, sq, ratio);
        }
        double coef = nj_new;
        for (int e = 0; e < isz - 1 - pos; ++e) coef *= Mfac;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            double pv = pcq[e];
            if (!KEEP_PC) pv = n < N ? (newest ? s_pc[n] : a.pc_arr[(size_t)slot * N + n]) : 0.0;
            if (n < N) {
                if (newest) a.pc_arr[(size_t)slot * N + n] = pv;
                if (rebuild && V_LDS) a.v_arr[(size_t)slot * N + n] = av[e];
            }
            if (n < a.ld) {
                a.V_f[(size_t)pos * a.ld + n] = (n < N) ? (float)av[e] : 0.f;
                a.B_f[(size_t)pos * a.ld + n] = (n < N) ? (float)(coef * pv) : 0.f;
            }
        }
    }
    if (tid == stid) {
        o.s = new_s;
        o.sigma = new_sigma;
        o.best_f = new_best;
        o.counteval = new_counteval;
        o.itr = itr + 1;
        o.iterator_sz = isz;
        o.imin_last = imin;
        o.done = new_sigma < 1e-20 ? 1 : 0;  // L/:426-429
        a.sc->sigma = (float)new_sigma;
        a.sc->mk = (float)mk;
        a.sc->k = isz;
        a.sc->generation = (uint32_t)(itr + 1);
        *a.psr_accum = 0ull;  // ready for the next generation's ranking
    }
    LMCMA_STAMP(5);
}

// ---- the update in two parts ----------------------------------------------------------
// Of the vectors rebuilt in a generation only the newest one (the last position of the
// time-ordered list, L/:365-366) starts from the new evolution path, and with it from this
// generation's costs.  The others start from paths stored in earlier generations and pass
// through transforms of vectors before them in the list: their rebuild needs nothing of
// this generation and can run beside its sampling and ranking (update_pre_body, one
// workgroup, e.g. an extra workgroup of the ranking launch).  What remains after the ranking
// (update_post_body) is the mean, the path, the step size, and the newest vector alone
// walking through the finished ones: no barrier per step.
//
// Both parts derive the slot bookkeeping (L/:331-364) from the tables of the last update;
// the second part writes the new tables.  The first part writes v_arr, nj and lj of the
// slots it rebuilds, nothing the sampling or ranking kernels read.

struct SlotPlan {
    int t_reg, vec_reg;  // the new tables, one entry per lane
    int imin, isz, newidx;
    int pos;             // position of this group's slot in the new list, -1: not in use
};

// t_reg / vec_reg: entry wl of the old tables in lane wl (m <= 64); grp: the slot of this 16-lane group
__device__ __forceinline__ SlotPlan slot_plan(int t_reg, int vec_reg, int itr, int m, int tid)
{
    const int wl = tid & 63, grp = tid >> 4;
    int imin = 1;
    if (itr < m) {
        if (wl == itr) t_reg = itr;
    } else if (m > 1) {
        const int vt = __shfl(vec_reg, t_reg, 64);  // vec[t[j]] in lane j
        const int vnext = __shfl_down(vt, 1, 64);
        const int gap = (wl < m - 1) ? vnext - vt : 0x7fffffff;
        const int dmin = wave_min_i32(gap);
        // first position with the smallest gap (the reference replaces on strict "<" only)
        imin = __ffsll((unsigned long long)__ballot(gap == dmin));
        if (dmin >= m) imin = 0;  // maxsteps = nvectors (L/:267)
        if (imin != m - 1) {
            const int sav = __shfl(t_reg, imin, 64);
            const int nxt = __shfl_down(t_reg, 1, 64);
            if (wl >= imin && wl < m - 1) t_reg = nxt;
            if (wl == m - 1) t_reg = sav;
        }
    } else {
        imin = 0;
    }
    const int isz = itr + 1 < m ? itr + 1 : m;
    const int newidx = __shfl(t_reg, isz - 1, 64);
    if (wl == newidx) vec_reg = itr;
    if (imin == 1) imin = 0;  // L/:373-374
    int pos = -1;
    const int g0 = (tid >> 6) << 2;  // first group of this wave
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long hit = __ballot(wl < isz && t_reg == g0 + q);
        if ((grp & 3) == q && hit != 0ull) pos = __ffsll(hit) - 1;
    }
    return {t_reg, vec_reg, imin, isz, newidx, pos};
}

// LDS doubles of the two parts (vectors by position, padded rows; norms and lj by position)
__host__ __device__ inline size_t update_pre_lds_doubles(int N, int m)
{
    const size_t np = 16 * (size_t)update_emax(N);
    return (size_t)m * np + 2 * LMCMA_HIP_MAX_VECTORS;
}

// First part.  T >= 16 m threads work (the workgroup may be larger: the others only meet the
// barriers).  EMAX >= N / 16.
template <int EMAX>
__device__ __forceinline__ void update_pre_body(const UpdateArgs &a, double *s_dyn, int T)
{
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    // Waves without vector work leave at once: a wave that has ended no longer counts at the
    // workgroup's barriers, which the loop below crosses once per step.
    if (tid >= T) return;
    const int N = a.N, m = a.m;
    const OptScalars &o = *a.opt;
    const int done = o.done;  // looked at after the loads have been requested (see update_body)
    const double c1 = o.c1, K = o.K;
    const int itr = o.itr;
    const bool worker = true;
    constexpr int NP = EMAX * 16;
    double *s_v = s_dyn;                         // [m][NP] by position, zero beyond N
    double *s_nv = s_v + (size_t)m * NP;         // [64] |v_i|^2
    double *s_ljp = s_nv + LMCMA_HIP_MAX_VECTORS;  // [64] lj of the untouched vectors

    int t_reg = 0, vec_reg = 0;
    if (wl < m) {
        t_reg = a.t[wl];
        vec_reg = a.vec[wl];
    }
    const bool has_slot = worker && grp < m;
    const int slot = has_slot ? grp : 0;
    const double lj_old = a.lj[slot];
    double vold[EMAX], pcq[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) pcq[e] = vold[e] = 0.0;
    if (worker) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            const size_t at = (size_t)slot * N + (n < N ? n : N - 1);
            pcq[e] = a.pc_arr[at];
            vold[e] = a.v_arr[at];
        }
    }
    asm volatile("" ::: "memory");
    if (done) return;  // nothing is rebuilt once the stop rule has fired
    const SlotPlan sp = slot_plan(t_reg, vec_reg, itr, m, tid);
    const int imin = sp.imin, isz = sp.isz, pos = sp.pos;
    const bool own = has_slot && pos >= 0;
    const bool newest = own && slot == sp.newidx;
    const bool rebuild = own && pos >= imin && !newest;
    const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1);

    double av[EMAX];
    double aa = 0.0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int n = e * 16 + lane;
        const bool in = own && n < N;
        av[e] = in ? (rebuild ? pcq[e] : vold[e]) : 0.0;  // L/:376-377: a rebuilt vector starts from its path
        aa += av[e] * av[e];
    }
    aa = row16_sum(aa);
    if (own && !rebuild && !newest) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) s_v[(size_t)pos * NP + e * 16 + lane] = av[e];
        if (lane == 0) {
            s_nv[pos] = aa;
            s_ljp[pos] = lj_old;
        }
    }
    __syncthreads();
    // vector i is final once it has received T_0 .. T_{i-1}; the last one rebuilt here is at
    // position isz - 2 and needs transforms up to isz - 3
    for (int i = 0; i < isz - 2; ++i) {
        const bool fresh = i >= imin;  // same for every thread
        if (fresh) {
            if (pos == i && rebuild) {
                double q0 = 0.0, q1 = 0.0;
#pragma unroll
                for (int e = 0; e < EMAX; e += 2) {
                    s_v[(size_t)i * NP + e * 16 + lane] = av[e];
                    s_v[(size_t)i * NP + (e + 1) * 16 + lane] = av[e + 1];
                    q0 += av[e] * av[e];
                    q1 += av[e + 1] * av[e + 1];
                }
                aa = row16_sum(q0 + q1);  // summed anew, not carried through the transforms (see update_body)
                if (lane == 0) s_nv[i] = aa;
            }
            __syncthreads();
        }
        if (rebuild && pos > i) {
            const double *vi = s_v + (size_t)i * NP;
            double vv[EMAX];
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int e = 0; e < EMAX; e += 2) {
                vv[e] = vi[e * 16 + lane];
                vv[e + 1] = vi[(e + 1) * 16 + lane];
                p0 += vv[e] * av[e];
                p1 += vv[e + 1] * av[e + 1];
            }
            const double nvi = s_nv[i];
            const double lji = fresh ? lj_of(nvi, sq, ratio) : s_ljp[i];
            const double p = row16_sum(p0 + p1);
            const double d = lji * p;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) av[e] = K * av[e] - d * vv[e];
        }
    }
    if (rebuild) {
        // L/:386-389; the last vector rebuilt here (position isz - 2) was not published above
        if (pos >= isz - 2) {
            double q = 0.0;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) q += av[e] * av[e];
            aa = row16_sum(q);
        }
        if (lane == 0) {
            a.nj[slot] = (sq / aa) * (sqrt(1 + ratio * aa) - 1);
            a.lj[slot] = lj_of(aa, sq, ratio);
        }
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            if (n < N) a.v_arr[(size_t)slot * N + n] = av[e];
        }
    }
}

// First part for a workgroup of 256 threads and up to 32 vectors: every 16-lane group owns two
// slots (grp and grp + 16), so each finished vector is read from LDS once for both and half as
// many waves meet at the barriers.  This is the form that rides in the sampling launch, whose
// workgroups have 256 threads.  EMAX >= N / 16.
template <int EMAX>
__device__ __forceinline__ void update_pre_body2(const UpdateArgs &a, double *s_dyn)
{
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    const int N = a.N, m = a.m;
    const OptScalars &o = *a.opt;
    const int done = o.done;  // looked at after the loads have been requested (see update_body)
    const double c1 = o.c1, K = o.K;
    const int itr = o.itr;
    constexpr int NP = EMAX * 16;
    double *s_v = s_dyn;                           // [m][NP] by position, zero beyond N
    double *s_nv = s_v + (size_t)m * NP;           // [64] |v_i|^2
    double *s_ljp = s_nv + LMCMA_HIP_MAX_VECTORS;  // [64] lj of the untouched vectors

    LMCMA_STAMP(6);
    int t_reg = 0, vec_reg = 0;
    if (wl < m) {
        t_reg = a.t[wl];
        vec_reg = a.vec[wl];
    }
    bool has[2];
    int slot[2];
    double lj_old[2];
    double av[2][EMAX];
    {
        // every load before the slot tables are looked at: both the path and the old vector of
        // both slots (which of the two a vector starts from is known only after the tables)
        double vold[2][EMAX], pcq[2][EMAX];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            has[q] = grp + 16 * q < m;
            slot[q] = has[q] ? grp + 16 * q : 0;
            lj_old[q] = a.lj[slot[q]];
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const int n = e * 16 + lane;
                const size_t at = (size_t)slot[q] * N + (n < N ? n : N - 1);
                pcq[q][e] = a.pc_arr[at];
                vold[q][e] = a.v_arr[at];
            }
        }
        asm volatile("" ::: "memory");
        if (done) return;  // nothing is rebuilt once the stop rule has fired
        LMCMA_STAMP(7);
        const SlotPlan sp = slot_plan(t_reg, vec_reg, itr, m, tid);
        t_reg = sp.t_reg;
        int pos[2] = {sp.pos, -1};
        {
            const int g0 = (tid >> 6) << 2;  // first group of this wave
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long hit = __ballot(wl < sp.isz && t_reg == g0 + q + 16);
                if ((grp & 3) == q && hit != 0ull) pos[1] = __ffsll(hit) - 1;
            }
        }
        const int imin = sp.imin, isz = sp.isz;
        const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1);
        bool own[2], rebuild[2];
        double nrm[2] = {0.0, 0.0};  // squared norm of a rebuilt vector, from its publication
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            own[q] = has[q] && pos[q] >= 0;
            const bool newest = own[q] && slot[q] == sp.newidx;
            rebuild[q] = own[q] && pos[q] >= imin && !newest;
            double aa = 0.0;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const int n = e * 16 + lane;
                const bool in = own[q] && n < N;
                av[q][e] = in ? (rebuild[q] ? pcq[q][e] : vold[q][e]) : 0.0;  // L/:376-377
                aa += av[q][e] * av[q][e];
            }
            aa = row16_sum(aa);
            if (own[q] && !rebuild[q] && !newest) {
#pragma unroll
                for (int e = 0; e < EMAX; ++e) s_v[(size_t)pos[q] * NP + e * 16 + lane] = av[q][e];
                if (lane == 0) {
                    s_nv[pos[q]] = aa;
                    s_ljp[pos[q]] = lj_old[q];
                }
            }
        }
        __syncthreads();
        LMCMA_STAMP(13);
        // as update_pre_body: vector i is final once it has received T_0 .. T_{i-1}
        for (int i = 0; i < isz - 2; ++i) {
            const bool fresh = i >= imin;  // same for every thread
            if (fresh) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (pos[q] == i && rebuild[q]) {
                        double q0 = 0.0, q1 = 0.0;
#pragma unroll
                        for (int e = 0; e < EMAX; e += 2) {
                            s_v[(size_t)i * NP + e * 16 + lane] = av[q][e];
                            s_v[(size_t)i * NP + (e + 1) * 16 + lane] = av[q][e + 1];
                            q0 += av[q][e] * av[q][e];
                            q1 += av[q][e + 1] * av[q][e + 1];
                        }
                        nrm[q] = row16_sum(q0 + q1);
                        if (lane == 0) s_nv[i] = nrm[q];
                    }
                }
                __syncthreads();
            }
            const bool need0 = rebuild[0] && pos[0] > i, need1 = rebuild[1] && pos[1] > i;
            if (need0 || need1) {
                const double *vi = s_v + (size_t)i * NP;
                double vv[EMAX];
                double p00 = 0.0, p01 = 0.0, p10 = 0.0, p11 = 0.0;
#pragma unroll
                for (int e = 0; e < EMAX; e += 2) {
                    vv[e] = vi[e * 16 + lane];
                    vv[e + 1] = vi[(e + 1) * 16 + lane];
                    p00 += vv[e] * av[0][e];
                    p01 += vv[e + 1] * av[0][e + 1];
                    p10 += vv[e] * av[1][e];
                    p11 += vv[e + 1] * av[1][e + 1];
                }
                const double nvi = s_nv[i];
                const double lji = fresh ? lj_of(nvi, sq, ratio) : s_ljp[i];
                const double d0 = lji * row16_sum(p00 + p01);
                const double d1 = lji * row16_sum(p10 + p11);
                if (need0) {
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) av[0][e] = K * av[0][e] - d0 * vv[e];
                }
                if (need1) {
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) av[1][e] = K * av[1][e] - d1 * vv[e];
                }
            }
        }
        LMCMA_STAMP(14);
#ifdef LMCMA_DIAG
        if (threadIdx.x == 0) g_update_stamps[15] = (unsigned long long)((imin << 8) | isz);
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (rebuild[q]) {
                // L/:386-389; the last vector rebuilt here (position isz - 2) was not published above
                double aa = nrm[q];
                if (pos[q] >= isz - 2) {
                    double q0 = 0.0, q1 = 0.0;
#pragma unroll
                    for (int e = 0; e < EMAX; e += 2) {
                        q0 += av[q][e] * av[q][e];
                        q1 += av[q][e + 1] * av[q][e + 1];
                    }
                    aa = row16_sum(q0 + q1);
                }
                if (lane == 0) {
                    a.nj[slot[q]] = (sq / aa) * (sqrt(1 + ratio * aa) - 1);
                    a.lj[slot[q]] = lj_of(aa, sq, ratio);
                }
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    const int n = e * 16 + lane;
                    if (n < N) a.v_arr[(size_t)slot[q] * N + n] = av[q][e];
                }
            }
        }
    }
}

// LDS doubles of the second part when TR threads read the partial rows
__host__ __device__ inline size_t update_post_lds_doubles(int N, int m, int TR)
{
    const size_t np = 16 * (size_t)update_emax(N);
    return (size_t)m * np + np + row_sum_lds_doubles(N, TR) + LMCMA_HIP_MAX_VECTORS;
}

// What the second part leaves in LDS when it runs at the head of the next generation's sampling
// launch, in EVERY workgroup of it (update_post_body<EMAX, true>): the float images the sampling
// reads, by position, and the sampling scalars.  Only the workgroup with write_state stores
// the state to memory.
struct FusedOut {
    float *sV, *sB;   // [rows][ldp], zeroed by the caller before the call; rows from k on stay zero
    float *s_xmean;   // [ldp]
    float *s_scal;    // [4]: sigma, M^k, k (bits of an int), generation (bits of a uint32)
    int ldp;
    bool write_state;
    // The storing workgroup overwrites state the others read at their start: every other
    // workgroup reports here once all its loads have arrived, and the storing one waits for
    // `others` reports before its first store (then puts the word back to zero).
    unsigned *counter;
    unsigned others;
};

struct NoHook {
    __device__ __forceinline__ void operator()(uint32_t) const {}
};

// Second part; thread roles as in update_body (T workers, TR row readers, the first wave
// past the workers evaluates the step-size rule).  Returns false when the stop rule had fired
// and nothing was done.  FUSED: see FusedOut; hook(generation) is called once the scalars of the
// state have arrived and the partial rows are still on their way (work that needs no more than
// the number of the generation about to be sampled).
// phase stamps of the second part (diagnostic build): in the fused form by the launch's last workgroup only
#ifdef LMCMA_DIAG
#define POST_STAMP(i)                                                                               \
    do {                                                                                            \
        if (threadIdx.x == 0 && (!FUSED || blockIdx.x == gridDim.x - 1))                            \
            g_update_stamps[i] = __builtin_amdgcn_s_memrealtime();                                  \
    } while (0)
#else
#define POST_STAMP(i)
#endif
template <int EMAX, bool FUSED = false, typename Hook = NoHook>
__device__ __forceinline__ bool update_post_body(const UpdateArgs &a, double *s_dyn, int T, int TR,
                                                 const FusedOut *fo = nullptr, Hook hook = NoHook())
{
    const bool ws = !FUSED || fo->write_state;  // this workgroup stores the state
    constexpr bool KEEP_PC = EMAX <= 16;  // the paths stay in registers (re-read at the end otherwise)
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wl = tid & 63;
    const int N = a.N, m = a.m, lambda = a.lambda;
    OptScalars &o = *a.opt;
    const int done = o.done;  // looked at after the loads have been requested (see update_body)
    const double sigma = o.sigma, cc = o.cc, c1 = o.c1, K = o.K, Mfac = o.M;
    const int itr = o.itr;
    const bool worker = tid < T;
    const int stid = (int)blockDim.x > T ? T : 0;
    const bool swave = (tid >> 6) == (stid >> 6);
    const double o_mueff = o.mueff, o_s = o.s, o_cs = o.cs, o_val_target = o.val_target, o_best_f = o.best_f;
    const long long o_counteval = o.counteval;

    constexpr int NP = EMAX * 16;
    double *s_v = s_dyn;                         // [m][NP] finished vectors by position
    double *s_pc = s_v + (size_t)m * NP;         // [NP] new evolution path
    double *s_red = s_pc + NP;                   // [rg][N]
    double *s_ljp = s_red + row_sum_lds_doubles(N, TR);  // [64] lj by position

    POST_STAMP(0);
    int t_reg = 0, vec_reg = 0;
    if (wl < m) {
        t_reg = a.t[wl];
        vec_reg = a.vec[wl];
    }
    const bool has_slot = worker && grp < m;
    const int slot = has_slot ? grp : 0;
    const double lj_old = a.lj[slot], nj_old = a.nj[slot];
    double xold[2] = {0.0, 0.0}, pcold[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = tid + q * T;
        if (worker && n < N) {
            xold[q] = a.xmean[n];
            pcold[q] = a.pc[n];
        }
    }
    uint32_t pp[16];
    unsigned long long psr_lane = 0ull;
    float best_in = 0.f;
    if (swave) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = wl + 64 * u;
            pp[u] = a.psr_parts[b < a.n_psr_parts ? b : 0];
        }
        if (a.scalars_in_rows) {
            if (wl < a.n_partials) psr_lane = (unsigned long long)a.partials[(size_t)wl * a.partial_stride + N];
            if (tid == stid) best_in = (float)a.partials[N + 1];
        } else if (tid == stid) {
            psr_lane = *a.psr_accum;
            best_in = a.sorted_cost[0];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) pp[u] = 0u;
    }
    double vold[EMAX], pcq[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) pcq[e] = vold[e] = 0.0;
    if (worker) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            const size_t at = (size_t)slot * N + (n < N ? n : N - 1);
            pcq[e] = a.pc_arr[at];
            vold[e] = a.v_arr[at];
        }
    }
    RowSumBatch first_rows;
    row_sums_begin(a.partials, a.n_partials, N, a.partial_stride, TR, first_rows);
    asm volatile("" ::: "memory");
    if (done) return false;  // the stop rule has fired: the state stays as it is
    POST_STAMP(1);
    hook((uint32_t)(itr + 1));

    const SlotPlan sp = slot_plan(t_reg, vec_reg, itr, m, tid);
    t_reg = sp.t_reg;
    vec_reg = sp.vec_reg;
    const int imin = sp.imin, isz = sp.isz, newidx = sp.newidx, pos = sp.pos;
    const bool own = has_slot && pos >= 0;
    const bool newest = own && slot == newidx;

#pragma unroll
    for (int u = 0; u < 16; ++u) psr_lane += (!a.scalars_in_rows && wl + 64 * u < a.n_psr_parts) ? pp[u] : 0u;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) psr_lane += __shfl_xor(psr_lane, sft, 64);

    const double pc_gain = sqrt(cc * (2 - cc) * o_mueff);
    const double sq = sqrt(1 - c1), ratio = c1 / (1 - c1);

    // L/:393-419 and the feedback bookkeeping, by one thread (see update_body)
    double new_sigma = sigma, new_s = 0.0, new_best = 0.0, mk = 1.0;
    long long new_counteval = 0;
    if (tid == stid) {
        new_s = o_s;
        if (itr > 0) {
            const long long cur_sum = (long long)lambda * (lambda - 1) / 2 + (long long)psr_lane;
            const long long total = (long long)lambda * (2ll * lambda - 1);
            const double mean_current = (double)cur_sum / lambda;
            const double mean_previous = (double)(total - cur_sum) / lambda;
            const double diffv = (mean_previous - mean_current) / lambda;
            const double z1 = diffv - o_val_target;
            new_s = (1 - o_cs) * new_s + o_cs * z1;
            new_sigma = sigma * exp(new_s / 1.0);
        }
        const double best = (double)best_in;
        new_counteval = o_counteval;
        new_best = (best < o_best_f || new_counteval == 0) ? best : o_best_f;
        new_counteval += lambda;
        for (int q = 0; q < isz; ++q) mk *= Mfac;
    }

    // every vector but the newest is final: publish it by position
    double av[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int n = e * 16 + lane;
        const bool in = own && n < N;
        if (!in) pcq[e] = 0.0;
        av[e] = in ? vold[e] : 0.0;
    }
    if (own && !newest) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) s_v[(size_t)pos * NP + e * 16 + lane] = av[e];
        if (lane == 0) s_ljp[pos] = lj_old;
    }
    row_sums_to_lds_t<true>(a.partials, a.n_partials, N, a.partial_stride, TR, s_red, first_rows);
    // Workgroup 0 overwrites the state once every other workgroup has reported below, so every load
    // of the old state issued above (mean, path, vectors: some are first used after this point) must
    // have returned before the report: a barrier alone does not wait for outstanding global loads.
    if (FUSED && !ws) __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    POST_STAMP(2);
#ifdef LMCMA_DIAG
    if (FUSED && !ws && tid == 0 && !(g_diag_skip_report != 0 && blockIdx.x == 1))
#else
    if (FUSED && !ws && tid == 0)
#endif
        __hip_atomic_fetch_add(fo->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // L/:316-329: new mean, evolution path
    double xnew[2] = {0.0, 0.0}, pcn[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = tid + q * T;
        if (worker && n < N) {
            xnew[q] = row_sums_total(s_red, N, n, row_sum_groups(N, TR));
            pcn[q] = (1 - cc) * pcold[q] + pc_gain * (xnew[q] - xold[q]) / sigma;
            s_pc[n] = pcn[q];
        }
    }
    for (int n = N + tid; n < NP; n += (int)blockDim.x) s_pc[n] = 0.0;
    __syncthreads();
    POST_STAMP(3);

    // L/:375-390 with invAz (:449-463) for the newest vector: v = T_{k-2} ... T_0 pc.  Its steps
    // depend on each other, and a step is bound by the issue of its fp64 instructions, so the
    // whole wave that holds the newest slot's group walks the chain, 64 lanes wide (coordinate
    // n = 64 j + lane; both operands come from LDS, so the lanes' other roles do not matter):
    // EMAX / 4 multiply-adds per lane and step where one 16-lane group needs EMAX.  The wave
    // also stores everything of the newest slot.
    constexpr int EW = (NP + 63) / 64;
    const bool chain_wave = worker && (tid >> 6) == (newidx >> 2);
    double avw[EW], pcw[EW];
    double nj_w = 0.0, lj_w = 0.0;
#pragma unroll
    for (int j = 0; j < EW; ++j) avw[j] = pcw[j] = 0.0;
    if (chain_wave) {
#ifdef LMCMA_DIAG
        if (wl == 0 && (!FUSED || blockIdx.x == gridDim.x - 1)) g_update_stamps[6] = __builtin_amdgcn_s_memrealtime();
#endif
        double vv[EW];
#pragma unroll
        for (int j = 0; j < EW; ++j) {
            const int n = j * 64 + wl;
            const bool in = EW * 64 == NP || n < NP;
            avw[j] = pcw[j] = in ? s_pc[in ? n : 0] : 0.0;
            vv[j] = in ? s_v[in ? n : 0] : 0.0;
        }
        double lji = s_ljp[0];
        for (int i = 0; i < isz - 1; ++i) {
            // the next vector and its lj are requested before this step's sums are under way
            const int nx = i + 1 < isz - 1 ? i + 1 : i;
            const double *vn = s_v + (size_t)nx * NP;
            double vnext[EW];
#pragma unroll
            for (int j = 0; j < EW; ++j) {
                const int n = j * 64 + wl;
                const bool in = EW * 64 == NP || n < NP;
                vnext[j] = in ? vn[in ? n : 0] : 0.0;
            }
            const double ljn = s_ljp[nx];
            asm volatile("" ::: "memory");  // (keeps the requests above this step's arithmetic)
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int j = 0; j < EW; j += 2) {
                p0 += vv[j] * avw[j];
                if (j + 1 < EW) p1 += vv[j + 1] * avw[j + 1];
            }
            const double d = lji * wave_sum(p0 + p1);
#pragma unroll
            for (int j = 0; j < EW; ++j) {
                avw[j] = K * avw[j] - d * vv[j];
                vv[j] = vnext[j];
            }
            lji = ljn;
        }
        double aa = 0.0;
#pragma unroll
        for (int j = 0; j < EW; ++j) aa += avw[j] * avw[j];
        aa = wave_sum(aa);
        nj_w = (sq / aa) * (sqrt(1 + ratio * aa) - 1);
        lj_w = lj_of(aa, sq, ratio);
#ifdef LMCMA_DIAG
        if (wl == 0 && (!FUSED || blockIdx.x == gridDim.x - 1)) g_update_stamps[7] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    POST_STAMP(4);

    // What the sampling reads, by position q: V_f[q] = v_q, B_f[q] = M^(k-1-q) nj_q pc_q.  FUSED:
    // into this workgroup's LDS (no address outside LDS is formed on this path).
    const bool older = own && !newest;
    float vf[EMAX], bf[EMAX];
    if (older) {
        double coef = nj_old;
        for (int e = 0; e < isz - 1 - pos; ++e) coef *= Mfac;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int n = e * 16 + lane;
            double pv = pcq[e];
            if (!KEEP_PC) pv = n < N ? a.pc_arr[(size_t)slot * N + n] : 0.0;
            vf[e] = (n < N) ? (float)av[e] : 0.f;
            bf[e] = (n < N) ? (float)(coef * pv) : 0.f;
        }
    }
    float vfw[EW], bfw[EW];
#pragma unroll
    for (int j = 0; j < EW; ++j) {
        const int n = j * 64 + wl;
        vfw[j] = (n < N) ? (float)avw[j] : 0.f;
        bfw[j] = (n < N) ? (float)(nj_w * pcw[j]) : 0.f;
    }
    if (FUSED) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = tid + q * T;
            if (worker && n < fo->ldp) fo->s_xmean[n] = (n < N) ? (float)xnew[q] : 0.f;
        }
        if (older) {
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const int n = e * 16 + lane;
                if (n < fo->ldp) {
                    fo->sV[pos * fo->ldp + n] = vf[e];
                    fo->sB[pos * fo->ldp + n] = bf[e];
                }
            }
        }
        if (chain_wave) {
#pragma unroll
            for (int j = 0; j < EW; ++j) {
                const int n = j * 64 + wl;
                if (n < NP && n < fo->ldp) {
                    fo->sV[(isz - 1) * fo->ldp + n] = vfw[j];
                    fo->sB[(isz - 1) * fo->ldp + n] = bfw[j];
                }
            }
        }
        if (tid == stid) {
            fo->s_scal[0] = (float)new_sigma;
            fo->s_scal[1] = (float)mk;
            fo->s_scal[2] = __int_as_float(isz);
            fo->s_scal[3] = __uint_as_float((uint32_t)(itr + 1));
        }
    }

    bool stores_ok = true;
    if (FUSED && ws && fo->others > 0) {
        // the other workgroups of the launch must have read the old state before it is overwritten
        __shared__ int s_go;
        if (tid == 0) {
            int spins = 0, ok = 1;
            while (__hip_atomic_load(fo->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fo->others) {
#ifdef LMCMA_DIAG
                if (++spins > (1 << 16)) {  // (the diagnostic build gives up after some tens of milliseconds)
#else
                if (++spins > (1 << 23)) {  // some seconds: the exit every wave reaches whatever the others do
#endif
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_go = ok;
        }
        __syncthreads();
        stores_ok = s_go != 0;
        if (tid == 0) {
            *fo->counter = 0u;
            if (!stores_ok) {
                o.done = 3;  // reported by the host as an error; the state is left as it was
                if (a.host_flag) __hip_atomic_store(a.host_flag, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }

    // the state back to memory, by the one workgroup that stores it
    if (ws && stores_ok) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = tid + q * T;
            if (worker && n < N) {
                a.xmean[n] = xnew[q];
                a.pc[n] = pcn[q];
            }
            if (worker && n < a.ld) a.xmean_f[n] = (n < N) ? (float)xnew[q] : 0.f;
        }
        if (tid < m) {
            a.t[tid] = t_reg;
            a.vec[tid] = vec_reg;
            if (tid < isz) a.iterator[tid] = t_reg;
        }
        if (older) {
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const int n = e * 16 + lane;
                if (n < a.ld) {
                    a.V_f[(size_t)pos * a.ld + n] = vf[e];
                    a.B_f[(size_t)pos * a.ld + n] = bf[e];
                }
            }
        }
        if (chain_wave) {
            if (wl == 0) {
                a.nj[newidx] = nj_w;
                a.lj[newidx] = lj_w;
            }
#pragma unroll
            for (int j = 0; j < EW; ++j) {
                const int n = j * 64 + wl;
                if (n < N) {
                    a.pc_arr[(size_t)newidx * N + n] = pcw[j];
                    a.v_arr[(size_t)newidx * N + n] = avw[j];
                }
                if (n < NP && n < a.ld) {
                    a.V_f[(size_t)(isz - 1) * a.ld + n] = vfw[j];
                    a.B_f[(size_t)(isz - 1) * a.ld + n] = bfw[j];
                }
            }
        }
        if (tid == stid) {
            o.s = new_s;
            o.sigma = new_sigma;
            o.best_f = new_best;
            o.counteval = new_counteval;
            o.itr = itr + 1;
            o.iterator_sz = isz;
            o.imin_last = imin;
            o.done = new_sigma < 1e-20 ? 1 : 0;  // L/:426-429
            a.sc->sigma = (float)new_sigma;
            a.sc->mk = (float)mk;
            a.sc->k = isz;
            a.sc->generation = (uint32_t)(itr + 1);
            *a.psr_accum = 0ull;
        }
    }
    POST_STAMP(5);
    return true;
}

#undef POST_STAMP

}  // namespace lmcma
