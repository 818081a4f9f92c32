// Elite recombination and the on-device LM-CMA update (fp64).
//
// Replaces LMCMA::update (lmcma_path_planner/src/lmcma.cpp:313-424) and invAz
// (:449-463); the citations L/:a-b below are lines of that file.
#include <stdlib.h>

#include "update_body.hpp"

namespace lmcma {

namespace {

// ---- weighted recombination, L/:321-326 --------------------------------------
// Block b owns ranks [32 b, 32 b + 32) of the elite list.  A thread owns one
// float4 column chunk of one "slot"; slot s walks ranks s, s + nslots, ... of the
// block in order, so all row loads of a block are independent and coalesced.
// The slots are then added in slot order through LDS, and k_update (or
// k_recombine_final) adds the blocks in block order: the summation tree is fixed
// by (mu, ld) alone.  Rows that live in another context are skipped.
__global__ __launch_bounds__(256) void k_recombine_partial(const float *rows, int ld,
                                                           int N, int first_candidate, int n_local,
                                                           const int32_t *perm,
                                                           const double *weights,
                                                           int mu, double *block_partials, size_t qstride)
{
    extern __shared__ __attribute__((aligned(16))) double s_part[];  // [nslots][ld]
    if (qstride != 0) {  // query blockIdx.y of a batch
        const size_t off = (size_t)blockIdx.y * qstride;
        shift_ptr(rows, off); shift_ptr(perm, off); shift_ptr(weights, off); shift_ptr(block_partials, off);
    }
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

// partial_local[n] = sum of the rows of block_partials, in the order of row_sums_to_lds.
// With x.best != nullptr the row gets two more entries, the scalars a context contributes to
// the update of a sharded population: [N] its success-rule count, [N + 1] the lowest cost.
__global__ __launch_bounds__(1024) void k_recombine_final(const double *__restrict__ block_partials,
                                                          int nblocks, int N,
                                                          double *__restrict__ partial_local, const FinalExtras x)
{
    __shared__ double s_red[2048];  // row_sum_lds_doubles(N <= 512, T <= 1024)
    row_sums_to_lds(block_partials, nblocks, N, N, (int)blockDim.x, s_red);
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const double total = row_sums_total(s_red, N, n, row_sum_groups(N, (int)blockDim.x));
        partial_local[n] = total;
        for (int d = 0; d < x.send.n; ++d) peer_store(static_cast<double *>(x.send.dst[d]) + n, total);
    }
    if (x.best != nullptr && threadIdx.x < 64) {
        unsigned long long v = 0ull;
        if (x.count_psr) {
            for (int b = threadIdx.x; b < x.n_parts; b += 64) v += x.psr_parts[b];
            if (threadIdx.x == 0 && x.psr_accum != nullptr) v += *x.psr_accum;
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
        if (threadIdx.x == 0) {
            const double best = (double)*x.best;
            partial_local[N] = (double)v;
            partial_local[N + 1] = best;
            for (int d = 0; d < x.send.n; ++d) {
                peer_store(static_cast<double *>(x.send.dst[d]) + N, (double)v);
                peer_store(static_cast<double *>(x.send.dst[d]) + N + 1, best);
            }
        }
    }
    if (x.send.n > 0) peer_arrive(x.send);
}

// The same sum by several workgroups, 16 columns each, for a row that leaves the context (a shard's
// contribution to the exchange; nothing else has to reproduce its summation order): one workgroup
// reads the freshly written rows at some tens of GB/s, N / 16 of them read a 128-byte strip each.
// Thread (g, c) adds rows g, g + 32, ... of columns 16 b + 2 c, + 1; the 32 partial sums of a
// column are then added in group order.  Even N.
__global__ __launch_bounds__(256) void k_recombine_final_wide(const double *block_partials, int nblocks, int N,
                                                              double *partial_local, const FinalExtras x,
                                                              size_t qstride)
{
    __shared__ double s_red[32][16];
    shift_ptr(block_partials, blockIdx.y * qstride);  // query blockIdx.y of a batch
    shift_ptr(partial_local, blockIdx.y * qstride);
    const int tid = threadIdx.x, g = tid >> 3, c = tid & 7;
    const int n = 16 * (int)blockIdx.x + 2 * c;
    double a0 = 0.0, a1 = 0.0;
    if (n < N) {
        for (int r0 = g; r0 < nblocks; r0 += 32 * 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 32 * u;
                v[u] = make_double2(0.0, 0.0);
                if (r < nblocks) v[u] = *reinterpret_cast<const double2 *>(block_partials + (size_t)r * N + n);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + 32 * u < nblocks) {
                    a0 += v[u].x;
                    a1 += v[u].y;
                }
        }
    }
    s_red[g][2 * c] = a0;
    s_red[g][2 * c + 1] = a1;
    __syncthreads();
    if (tid < 16 && 16 * (int)blockIdx.x + tid < N) {
        double total = 0.0;
        for (int q = 0; q < 32; ++q) total += s_red[q][tid];
        partial_local[16 * blockIdx.x + tid] = total;
        for (int d = 0; d < x.send.n; ++d) peer_store(static_cast<double *>(x.send.dst[d]) + 16 * blockIdx.x + tid, total);
    }
    if (blockIdx.x == 0 && x.best != nullptr && tid >= 64 && tid < 128) {
        const int l = tid - 64;
        unsigned long long v = 0ull;
        if (x.count_psr) {
            for (int b = l; b < x.n_parts; b += 64) v += x.psr_parts[b];
            if (l == 0 && x.psr_accum != nullptr) v += *x.psr_accum;
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
        if (l == 0) {
            const double best = (double)*x.best;
            partial_local[N] = (double)v;
            partial_local[N + 1] = best;
            for (int d = 0; d < x.send.n; ++d) {
                peer_store(static_cast<double *>(x.send.dst[d]) + N, (double)v);
                peer_store(static_cast<double *>(x.send.dst[d]) + N + 1, best);
            }
        }
    }
    if (x.send.n > 0) peer_arrive(x.send);  // (uniform over the launch)
    if (x.gather.flags != nullptr) {        // (likewise)
        peer_wait(x.gather);
        for (int q = blockIdx.x * 256 + tid; q < x.gather_quads; q += gridDim.x * 256) x.gather_dst[q] = x.gather_src[q];
    }
}

// A first level for many partial rows: rows [b chunk, (b + 1) chunk) of src -> row b of dst.
// One workgroup reads freshly written rows at some tens of GB/s, so beyond a hundred rows or so
// the single workgroup that consumes them is better off with a few dozen rows summed by
// many workgroups first.
__global__ __launch_bounds__(256) void k_reduce_rows(const double *src, int rows, int chunk, int N,
                                                     double *dst, size_t qstride)
{
    __shared__ double s_red[1024];  // row_sum_lds_doubles(N <= 512, 256)
    shift_ptr(src, blockIdx.y * qstride);
    shift_ptr(dst, blockIdx.y * qstride);
    const int first = blockIdx.x * chunk;
    const int count = rows - first < chunk ? rows - first : chunk;
    row_sums_to_lds(src + (size_t)first * N, count, N, N, 256, s_red);
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256)
        dst[(size_t)blockIdx.x * N + n] = row_sums_total(s_red, N, n, row_sum_groups(N, 256));
}

// ---- update, stand-alone ------------------------------------------------------------
// TB threads do the vector work.  The workgroup is larger when there is room: 512 threads
// for TB = 256 (two waves per SIMD keep the whole register file per lane), all of which
// read the partial rows and the first extra wave evaluates the step-size rule.  A ninth
// wave would cut the registers per lane to a third of the file, and the kernel spills.
constexpr int update_block_threads(int TB) { return TB == 256 ? 512 : TB; }
constexpr int update_row_threads(int TB) { return TB == 256 ? 512 : TB; }

template <int TB, int EMAX, bool V_LDS>
__global__ __launch_bounds__(update_block_threads(TB)) void k_update(const UpdateArgs a)
{
    // all LDS in one dynamic array of doubles
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    update_body<EMAX, V_LDS>(shifted(a, blockIdx.y), s_dyn, TB, update_row_threads(TB));
}

template <int TB, int EMAX>
__global__ __launch_bounds__(TB) void k_update_pre(const UpdateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    update_pre_body<EMAX>(shifted(a, blockIdx.y), s_dyn, TB);
}

template <int TB, int EMAX>
__global__ __launch_bounds__(update_block_threads(TB)) void k_update_post(const UpdateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    update_post_body<EMAX>(shifted(a, blockIdx.y), s_dyn, TB, update_row_threads(TB));
}

}  // namespace

#ifdef LMCMA_DIAG
extern "C" hipError_t lmcma_diag_update_stamps(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_update_stamps), 16 * sizeof(unsigned long long));
}
#endif

// workgroup size of the update for m direction vectors of N parameters: 16 lanes per
// vector; the smallest workgroup leaves the most registers per lane
static int threads_for_vectors(int m)
{
    int t = m <= 16 ? 256 : m <= 32 ? 512 : 1024;
    // (a measurement switch, read once: more threads than 16 m only help read the partial rows)
    static const int v = getenv("LMCMA_HIP_UPDATE_THREADS") ? atoi(getenv("LMCMA_HIP_UPDATE_THREADS")) : 0;
    if ((v == 256 || v == 512 || v == 1024) && v >= t) t = v;
    return t;
}

static bool update_fits_lds(int N, int m)
{
    return update_lds_doubles(N, m, update_row_threads(threads_for_vectors(m)), true) * sizeof(double) <= 150 * 1024;
}

// threads that add the partial rows in the update (the order of that sum depends on it)
int update_threads(int N, int m) { return update_fits_lds(N, m) ? update_row_threads(threads_for_vectors(m)) : 1024; }

bool update_is_split(int N, int m)
{
    static const bool off = getenv("LMCMA_HIP_UPDATE_ONE_KERNEL") != nullptr;
    // Up to 32 vectors.  With up to 16 the first part rides in the ranking launch; with more it
    // (30 dependent steps at m = 32) outlasts that launch by more than the second part saves
    // (measured 64.7 against 61.2 us per generation at lambda = 8192, N = 144, m = 32), so it rides
    // in the longer sampling launch instead (update_pre_rides_with_ranking, launch_generate).
    return !off && m >= 1 && threads_for_vectors(m) <= 512 && N >= 1 && N <= 512 && update_fits_lds(N, m);
}

bool update_pre_rides_with_ranking(int m) { return threads_for_vectors(m) == 256; }

int update_pre_threads(int m) { return threads_for_vectors(m); }

int recombine_num_blocks(int mu)
{
    return (mu + kRecombineRanksPerBlock - 1) / kRecombineRanksPerBlock;
}

hipError_t launch_recombine(hipStream_t s, const float *rows, int ld, int N, int first_candidate,
                            int n_local, const int32_t *perm, const double *weights, int mu,
                            double *block_partials, Batch batch)
{
    if (mu <= 0 || N <= 0 || (ld & 3) || ld > 512 || batch.queries < 1) return hipErrorInvalidValue;
    const int nb = recombine_num_blocks(mu);
    const int nslots = 256 / (ld >> 2);
    const size_t lds = (size_t)nslots * ld * sizeof(double);
    hipLaunchKernelGGL(k_recombine_partial, dim3(nb, batch.queries), dim3(256), lds, s, rows, ld, N, first_candidate,
                       n_local, perm, weights, mu, block_partials, batch.stride);
    return hipGetLastError();
}

// Many partial rows of even length go to ONE row (the strips of k_recombine_final_wide), so the
// update that follows reads a single row; odd lengths keep the first level of at most 32 rows.
static bool reduce_to_one_row(int N)
{
    static const bool off = getenv("LMCMA_HIP_REDUCE_TWO_LEVELS") != nullptr;
    return !off && (N & 1) == 0;
}

int reduce_rows_out(int rows, int N)
{
    if (reduce_to_one_row(N)) return 1;
    const int chunk = (rows + kReducedRowsMax - 1) / kReducedRowsMax;
    return (rows + chunk - 1) / chunk;
}

hipError_t launch_reduce_rows(hipStream_t s, const double *src, int rows, int N, double *dst, Batch batch)
{
    if (rows < 1 || N < 1 || N > 512 || batch.queries < 1) return hipErrorInvalidValue;
    if (reduce_to_one_row(N)) {
        hipLaunchKernelGGL(k_recombine_final_wide, dim3((N + 15) / 16, batch.queries), dim3(256), 0, s, src, rows, N, dst,
                           FinalExtras{}, batch.stride);
        return hipGetLastError();
    }
    const int chunk = (rows + kReducedRowsMax - 1) / kReducedRowsMax;
    hipLaunchKernelGGL(k_reduce_rows, dim3(reduce_rows_out(rows, N), batch.queries), dim3(256), 0, s, src, rows, chunk, N,
                       dst, batch.stride);
    return hipGetLastError();
}

bool recombine_final_is_wide(int rows, int N) { return (N & 1) == 0 && rows >= 32; }

hipError_t launch_recombine_final(hipStream_t s, const double *block_partials, int rows, int N,
                                  double *partial_local, int threads, const FinalExtras *extras, int *arrivals_out)
{
    if (rows < 1 || N < 1) return hipErrorInvalidValue;
    FinalExtras x{};
    if (extras) x = *extras;
    if (arrivals_out) *arrivals_out = 1;
    if (threads == 0) {  // the caller leaves the summation order to this function
        if (recombine_final_is_wide(rows, N)) {
            if (arrivals_out) *arrivals_out = (N + 15) / 16;
            hipLaunchKernelGGL(k_recombine_final_wide, dim3((N + 15) / 16), dim3(256), 0, s, block_partials, rows, N,
                               partial_local, x, (size_t)0);
            return hipGetLastError();
        }
        threads = 1024;
    }
    hipLaunchKernelGGL(k_recombine_final, dim3(1), dim3(threads), 0, s, block_partials, rows, N, partial_local, x);
    return hipGetLastError();
}

template <int TB, int EMAX, bool V_LDS>
static hipError_t launch_update_t(hipStream_t s, const UpdateArgs &a)
{
    const size_t lds = update_lds_doubles(a.N, a.m, update_row_threads(TB), V_LDS) * sizeof(double);
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_update<TB, EMAX, V_LDS>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_update<TB, EMAX, V_LDS>), dim3(1, a.batch.queries), dim3(update_block_threads(TB)), lds, s, a);
    return hipGetLastError();
}

template <int TB, int EMAX>
static hipError_t launch_pre_t(hipStream_t s, const UpdateArgs &a)
{
    const size_t lds = update_pre_lds_doubles(a.N, a.m) * sizeof(double);
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_update_pre<TB, EMAX>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_update_pre<TB, EMAX>), dim3(1, a.batch.queries), dim3(TB), lds, s, a);
    return hipGetLastError();
}

template <int TB, int EMAX>
static hipError_t launch_post_t(hipStream_t s, const UpdateArgs &a)
{
    const size_t lds = update_post_lds_doubles(a.N, a.m, update_row_threads(TB)) * sizeof(double);
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_update_post<TB, EMAX>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_update_post<TB, EMAX>), dim3(1, a.batch.queries), dim3(update_block_threads(TB)), lds, s, a);
    return hipGetLastError();
}

#define LMCMA_UPD_SPLIT(FN, TB)                              \
    do {                                                     \
        if (emax == 8) return FN<TB, 8>(s, a);               \
        if (emax == 10) return FN<TB, 10>(s, a);             \
        if (emax == 12) return FN<TB, 12>(s, a);             \
        if (emax == 16) return FN<TB, 16>(s, a);             \
        if (emax == 20) return FN<TB, 20>(s, a);             \
        if (emax == 24) return FN<TB, 24>(s, a);             \
        return FN<TB, 32>(s, a);                             \
    } while (0)

hipError_t launch_update_pre(hipStream_t s, const UpdateArgs &a)
{
    if (!update_is_split(a.N, a.m) || a.batch.queries < 1) return hipErrorInvalidValue;
    const int threads = threads_for_vectors(a.m), emax = update_emax(a.N);
    if (threads == 256) LMCMA_UPD_SPLIT(launch_pre_t, 256);
    if (threads == 512) LMCMA_UPD_SPLIT(launch_pre_t, 512);
    LMCMA_UPD_SPLIT(launch_pre_t, 1024);
}

hipError_t launch_update_post(hipStream_t s, const UpdateArgs &a)
{
    if (!update_is_split(a.N, a.m) || a.n_partials < 1 || a.batch.queries < 1) return hipErrorInvalidValue;
    const int threads = threads_for_vectors(a.m), emax = update_emax(a.N);
    if (threads == 256) LMCMA_UPD_SPLIT(launch_post_t, 256);
    if (threads == 512) LMCMA_UPD_SPLIT(launch_post_t, 512);
    LMCMA_UPD_SPLIT(launch_post_t, 1024);
}
#undef LMCMA_UPD_SPLIT

hipError_t launch_update(hipStream_t s, const UpdateArgs &a)
{
    if (a.m < 1 || a.m > LMCMA_HIP_MAX_VECTORS || a.N < 1 || a.N > 512 || a.n_partials < 1 || a.batch.queries < 1)
        return hipErrorInvalidValue;
    const int threads = threads_for_vectors(a.m);
    const int emax = update_emax(a.N);
    if (!update_fits_lds(a.N, a.m)) return launch_update_t<1024, 32, false>(s, a);  // m > 32 and N > 256
#define LMCMA_UPD(TB)                                          \
    do {                                                       \
        if (emax == 8) return launch_update_t<TB, 8, true>(s, a);   \
        if (emax == 10) return launch_update_t<TB, 10, true>(s, a); \
        if (emax == 12) return launch_update_t<TB, 12, true>(s, a); \
        if (emax == 16) return launch_update_t<TB, 16, true>(s, a); \
        if (emax == 20) return launch_update_t<TB, 20, true>(s, a); \
        if (emax == 24) return launch_update_t<TB, 24, true>(s, a); \
        return launch_update_t<TB, 32, true>(s, a);            \
    } while (0)
    if (threads == 256) LMCMA_UPD(256);
    if (threads == 512) LMCMA_UPD(512);
    LMCMA_UPD(1024);
#undef LMCMA_UPD
}

}  // namespace lmcma
