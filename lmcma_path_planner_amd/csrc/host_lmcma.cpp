// Host side of the optimiser: the LMCMA class of include/lmcma.hpp, its random
// number generator and the smoothness-matrix helpers.
//
// The class keeps the fp64 search state on the host and follows the order of
// operations of the reference's LMCMA::update (lmcma_path_planner/src/lmcma.cpp:313-424,
// cited below as L/:a-b); the population work goes to the GPU through lmcma_hip.h.
#include "lmcma.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmcma_hip.h"

// =============================================================================
// Random numbers: minimal-standard Lehmer generator (a = 16807, modulus 2^31-1,
// Schrage's factorisation q = 127773, r = 2836) behind a 32-entry shuffle table,
// and the polar method for normals.  L/:14-82.
// =============================================================================

namespace {

const long kLehmerA = 16807, kLehmerM = 2147483647, kLehmerQ = 127773, kLehmerR = 2836;

inline long lehmer_next(long s)
{
    const long hi = s / kLehmerQ;
    s = kLehmerA * (s - hi * kLehmerQ) - kLehmerR * hi;
    return s < 0 ? s + kLehmerM : s;
}

}  // namespace

void random_exit(random_t *t)
{
    delete[] t->rgrand;
    t->rgrand = 0;
}

long random_Start(random_t *t, long unsigned inseed)
{
    t->flgstored = 0;
    t->startseed = (long)inseed;
    if (inseed < 1) inseed = 1;
    long s = (long)inseed;
    // 40 warm-up draws; the last 32 fill the table from the top down (L/:23-30)
    for (int i = 39; i >= 0; --i) {
        s = lehmer_next(s);
        if (i < 32) t->rgrand[i] = s;
    }
    t->aktseed = s;
    t->aktrand = t->rgrand[0];
    return (long)inseed;
}

long random_init(random_t *t, long unsigned inseed)
{
    t->flgstored = 0;
    t->rgrand = new long[32];
    if (inseed < 1) {
        // a seed below 1 asks for a clock-derived one (L/:40-45)
        const clock_t c0 = clock();
        while (clock() == c0) {}
        inseed = (long unsigned)labs((long)(100.0 * (double)time(NULL) + (double)clock()));
    }
    return random_Start(t, inseed);
}

double random_Uniform(random_t *t)
{
    t->aktseed = lehmer_next(t->aktseed);
    const long slot = t->aktrand / 67108865;  // 2^31 / 32, rounded up (L/:57)
    t->aktrand = t->rgrand[slot];
    t->rgrand[slot] = t->aktseed;
    return (double)t->aktrand / 2.147483647e9;
}

double random_Gauss(random_t *t)
{
    if (t->flgstored) {
        t->flgstored = 0;
        return t->hold;
    }
    double a, b, r2;
    do {
        a = 2.0 * random_Uniform(t) - 1.0;
        b = 2.0 * random_Uniform(t) - 1.0;
        r2 = a * a + b * b;
    } while (r2 >= 1 || r2 <= 0);
    const double f = sqrt(-2.0 * log(r2) / r2);
    // the second coordinate goes out first, the first one is kept (L/:78-81)
    t->flgstored = 1;
    t->hold = f * a;
    return f * b;
}

// =============================================================================
// Sorting.  L/:84-104 sorts (value, id) pairs with libc qsort and a comparator
// that only looks at the value; with glibc that is a merge sort, so equal values
// stay in input order.  std::stable_sort states that contract explicitly.
// =============================================================================

int compare(const void *a, const void *b)
{
    const double x = ((const sortedvals *)a)->value, y = ((const sortedvals *)b)->value;
    return (x < y) ? -1 : ((x > y) ? 1 : 0);
}

void myqsort(int sz, double *arfitness, int *arindex, sortedvals *arr)
{
    for (int i = 0; i < sz; ++i) {
        arr[i].value = arfitness[i];
        arr[i].id = i;
    }
    std::stable_sort(arr, arr + sz, [](const sortedvals &p, const sortedvals &q) {
        // NaN is not ordered by the reference's comparator (undefined there); rank it last
        const bool pn = p.value != p.value, qn = q.value != q.value;
        if (pn || qn) return !pn && qn;
        return p.value < q.value;
    });
    for (int i = 0; i < sz; ++i) {
        arfitness[i] = arr[i].value;
        arindex[i] = arr[i].id;
    }
}

// =============================================================================
// Smoothness matrices.  L/:756-864.  The reference leaves the inverse and the
// Cholesky factor to Eigen (not vendored); the same factorisations are written
// out here.
// =============================================================================

void differentiationMatrix(int num_time_steps, int order, double dt, double *diff_matrix, int rowLen)
{
    // finite-difference rules of length 7, centred (L/:760-765)
    static const double rules[4][7] = {
        {0, 0, 0, 1, 0, 0, 0},
        {0, 0, -1, 1, 0, 0, 0},
        {0, -1 / 12.0, 16 / 12.0, -30 / 12.0, 16 / 12.0, -1 / 12.0, 0},
        {0, 1 / 12.0, -17 / 12.0, 46 / 12.0, -46 / 12.0, 17 / 12.0, -1 / 12.0}};
    if (rowLen < 0) rowLen = num_time_steps;
    const double scale = 1.0 / pow(dt, (int)order);
    for (int i = 0; i < num_time_steps; ++i) {
        double *row = diff_matrix + (size_t)i * rowLen;
        std::fill(row, row + num_time_steps, 0.0);
        // taps that fall off either end of the trajectory are dropped (L/:826-829)
        const int lo = std::max(0, i - 3), hi = std::min(num_time_steps - 1, i + 3);
        for (int c = lo; c <= hi; ++c) row[c] += scale * rules[order][c - i + 3];
    }
}

void invert(double *A, double *Ainv, int N)
{
    // Gauss-Jordan elimination with partial pivoting on [A | I]; A is read in
    // column-major order like the reference's Eigen::Map (L/:838).
    const int w = 2 * N;
    std::vector<double> aug((size_t)N * w, 0.0);
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) aug[(size_t)r * w + c] = A[(size_t)c * N + r];
        aug[(size_t)r * w + N + r] = 1.0;
    }
    for (int c = 0; c < N; ++c) {
        int p = c;
        for (int r = c + 1; r < N; ++r)
            if (fabs(aug[(size_t)r * w + c]) > fabs(aug[(size_t)p * w + c])) p = r;
        if (aug[(size_t)p * w + c] == 0.0) throw std::runtime_error("invert: singular matrix");
        if (p != c) std::swap_ranges(&aug[(size_t)c * w], &aug[(size_t)c * w] + w, &aug[(size_t)p * w]);
        const double d = aug[(size_t)c * w + c];
        for (int j = 0; j < w; ++j) aug[(size_t)c * w + j] /= d;
        for (int r = 0; r < N; ++r) {
            const double f = aug[(size_t)r * w + c];
            if (r == c || f == 0.0) continue;
            for (int j = 0; j < w; ++j) aug[(size_t)r * w + j] -= f * aug[(size_t)c * w + j];
        }
    }
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) Ainv[(size_t)c * N + r] = aug[(size_t)r * w + N + c];
}

void cholesky(double *C, double *L, int N)
{
    // lower factor, stored column-major: L[col * N + row] (L/:846-853)
    std::fill(L, L + (size_t)N * N, 0.0);
    for (int c = 0; c < N; ++c) {
        double d = C[(size_t)c * N + c];
        for (int k = 0; k < c; ++k) d -= L[(size_t)k * N + c] * L[(size_t)k * N + c];
        if (!(d > 0.0)) throw std::runtime_error("cholesky: matrix is not positive definite");
        d = sqrt(d);
        L[(size_t)c * N + c] = d;
        for (int r = c + 1; r < N; ++r) {
            double s = C[(size_t)c * N + r];
            for (int k = 0; k < c; ++k) s -= L[(size_t)k * N + r] * L[(size_t)k * N + c];
            L[(size_t)c * N + r] = s / d;
        }
    }
}

void applyCovL(double *L, double *z, int N)
{
    // z <- L z for the column-major lower factor (L/:857-864)
    std::vector<double> out(N, 0.0);
    for (int c = 0; c < N; ++c) {
        const double zc = z[c];
        const double *col = L + (size_t)c * N;
        for (int r = c; r < N; ++r) out[r] += col[r] * zc;
    }
    std::copy(out.begin(), out.end(), z);
}

void covariance(int num_dims, int num_waypoints, double *cov)
{
    // block-diagonal acceleration matrix A (one block per dimension, L/:786-791),
    // R = A A (L/:794-798), cov = R^-1 / (largest variance * waypoints) (L/:800-809)
    const int n = num_dims * num_waypoints;
    std::vector<double> A((size_t)n * n, 0.0), R((size_t)n * n, 0.0);
    for (int d = 0; d < num_dims; ++d)
        differentiationMatrix(num_waypoints, 2, 1.0, &A[(size_t)d * num_waypoints * n + (size_t)d * num_waypoints], n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * A[(size_t)k * n + j];
            R[(size_t)i * n + j] = s;
        }
    invert(R.data(), cov, n);
    double top = 0.0;
    for (int i = 0; i < n; ++i) top = std::max(top, cov[(size_t)i * n + i]);
    const double scaling = top * num_waypoints;
    for (size_t i = 0; i < (size_t)n * n; ++i) cov[i] /= scaling;
}

// =============================================================================
// LMCMA
// =============================================================================

struct LMCMA::Impl {
    // constructor arguments (pointers are the caller's, L/:109-110)
    double *initialParams, *loBounds, *hiBounds, *covarianceIn;
    int lambdaArg, inseed;
    bool verbose;
    // options
    int mArg = 0, device = 0;
    bool philox = false, deviceUpdate = false, haveProblem = false;
    int world = 1, rank = 0;      // setPopulationShard
    std::string uniqueId;
    LMCMATrajectoryProblem problem;
    std::vector<uint8_t> cellStore;

    int N = 0, lambda = 0, mu = 0, m = 0;
    double mueff = 0, sigma = 1.0, c1 = 0, cc = 0, cs = 0.3, target = 0.25, K = 1, M = 1, s = 0;
    int itr = 0, sampleIdx = 0, active = 0;
    std::vector<double> weights, xmean, xold, pc, pcArr, vArr, nj, lj, Lchol;
    std::vector<int> slot, born, order;   // t[], vec[], iterator[] of the reference
    std::vector<double> fitness, prevSorted, rows, lastCosts;
    bool rowsFresh = false;
    bool popReady = false;   // device update: the current generation's candidates have been drawn for ask/tell
    random_t rng;
    bool rngReady = false;
    lmcma_hip_ctx *ctx = 0;
    std::string error;
    bool initialised = false;

    void hip(int rc, const char *what)
    {
        if (rc == LMCMA_HIP_OK) return;
        error = std::string(what) + ": " + lmcma_hip_last_error();
        throw std::runtime_error(error);
    }

    void fetchRows()
    {
        if (rowsFresh) return;
        hip(lmcma_hip_get_rows(ctx, 0, lambda / world, rows.data()), "lmcma_hip_get_rows");
        rowsFresh = true;
    }
};

LMCMA::LMCMA(double *initialParams, int lambda, double *loBounds, double *hiBounds, double sigma,
             double *covariance, int inseed, bool verbose)
    : CMABase(initialParams, lambda, loBounds, hiBounds, covariance, inseed, verbose), impl(new Impl())
{
    impl->initialParams = initialParams;
    impl->lambdaArg = lambda;
    impl->loBounds = loBounds;
    impl->hiBounds = hiBounds;
    impl->sigma = sigma;
    impl->covarianceIn = covariance;
    impl->inseed = inseed;
    impl->verbose = verbose;
    // the generator is seeded at construction like CMABase's (L/:114)
    random_init(&impl->rng, (long unsigned)(inseed < 1 ? 0 : inseed));
    impl->rngReady = true;
}

LMCMA::~LMCMA()
{
    if (impl->ctx) lmcma_hip_destroy(impl->ctx);
    if (impl->rngReady) random_exit(&impl->rng);
    delete impl;
}

void LMCMA::setDirectionVectors(int m) { impl->mArg = m; }
void LMCMA::setDevice(int ordinal) { impl->device = ordinal; }
void LMCMA::usePhiloxSampling(bool on) { impl->philox = on; }
void LMCMA::useDeviceUpdate(bool on) { impl->deviceUpdate = on; }

void LMCMA::setPopulationShard(int world, int rank, const char *uniqueId)
{
    if (impl->initialised) throw std::runtime_error("setPopulationShard must come before init");
    if (world < 1 || rank < 0 || rank >= world || !uniqueId) throw std::invalid_argument("setPopulationShard: bad world / rank / id");
    impl->world = world;
    impl->rank = rank;
    impl->uniqueId.assign(uniqueId, uniqueId + LMCMA_HIP_UNIQUE_ID_BYTES);
}

void LMCMA::populationUniqueId(char uniqueId[128])
{
    if (lmcma_hip_comm_unique_id(uniqueId) != LMCMA_HIP_OK)
        throw std::runtime_error(std::string("lmcma_hip_comm_unique_id: ") + lmcma_hip_last_error());
}

void LMCMA::setTrajectoryProblem(const LMCMATrajectoryProblem &p)
{
    if (impl->initialised) throw std::runtime_error("setTrajectoryProblem must be called before init");
    if (!p.cells || p.waypoints < 1 || p.dims < 1 || p.dims > 3 || p.nx < 1 || p.ny < 1 || p.nz < 1)
        throw std::invalid_argument("setTrajectoryProblem: incomplete problem");
    impl->problem = p;
    const size_t cells = (size_t)p.nx * p.ny * p.nz;
    size_t bytes = cells;
    if (p.cell_type == LMCMA_GRID_BITS) bytes = ((cells + 31) / 32) * 4;
    else if (p.cell_type == LMCMA_GRID_F32) bytes = cells * 4;
    impl->cellStore.assign((const uint8_t *)p.cells, (const uint8_t *)p.cells + bytes);
    impl->problem.cells = impl->cellStore.data();
    impl->haveProblem = true;
}

void LMCMA::init(int N)
{
    Impl &d = *impl;
    if (d.initialised) throw std::runtime_error("LMCMA::init called twice");
    if (N < 1) throw std::invalid_argument("LMCMA::init: N must be positive");
    d.N = N;
    // population defaults, recombination weights, mueff: L/:134-156
    d.lambda = d.lambdaArg < 1 ? 4 + (int)(3 * log((double)N)) : d.lambdaArg;
    d.mu = d.lambda / 2;
    if (d.mu < 1) throw std::invalid_argument("LMCMA::init: lambda must be at least 2");
    d.weights.resize(d.mu);
    double wsum = 0;
    for (int i = 0; i < d.mu; ++i) {
        d.weights[i] = log((double)d.mu + 0.5) - log((double)(1 + i));
        wsum += d.weights[i];
    }
    d.mueff = 0;
    for (int i = 0; i < d.mu; ++i) {
        d.weights[i] /= wsum;
        d.mueff += d.weights[i] * d.weights[i];
    }
    d.mueff = 1.0 / d.mueff;
    d.xmean.resize(N);
    d.xold.assign(N, 0.0);
    if (d.initialParams) std::copy(d.initialParams, d.initialParams + N, d.xmean.begin());
    else for (int i = 0; i < N; ++i) d.xmean[i] = random_Uniform(&d.rng);  // L/:161-163
    if (d.covarianceIn) {
        d.Lchol.resize((size_t)N * N);
        cholesky(d.covarianceIn, d.Lchol.data(), N);  // L/:165-169
    }
    // LM-CMA constants: L/:266-272, with the number of direction vectors free
    d.m = d.mArg < 1 ? d.lambda : d.mArg;
    d.c1 = 1.0 / (10 * log((double)(N + 1)));
    d.cc = 1.0 / d.m;
    d.K = 1 / sqrt(1 - d.c1);
    d.M = sqrt(1 - d.c1);
    d.pc.assign(N, 0.0);
    d.pcArr.assign((size_t)d.m * N, 0.0);
    d.vArr.assign((size_t)d.m * N, 0.0);
    d.nj.assign(d.m, 0.0);
    d.lj.assign(d.m, 0.0);
    d.slot.assign(d.m, 0);
    d.born.assign(d.m, 0);
    d.order.assign(d.m, 0);
    d.fitness.assign(d.lambda, 0.0);
    d.prevSorted.assign(d.lambda, 0.0);
    d.lastCosts.assign(d.lambda, 0.0);
    d.rows.assign((size_t)d.lambda * N, 0.0);
    d.s = 0;
    d.active = 0;

    if (d.deviceUpdate && !d.philox)
        throw std::invalid_argument("useDeviceUpdate needs usePhiloxSampling (the device draws its own normals)");
    if (d.haveProblem && d.problem.dims * d.problem.waypoints != N)
        throw std::invalid_argument("LMCMA::init: N must equal dims * waypoints of the trajectory problem");

    lmcma_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = d.device;
    cfg.dims = d.haveProblem ? d.problem.dims : 0;
    cfg.waypoints = d.haveProblem ? d.problem.waypoints : 0;
    cfg.n_params = N;
    const bool sharded = !d.uniqueId.empty();
    if (sharded && !(d.deviceUpdate && d.lambda % d.world == 0))
        throw std::invalid_argument("setPopulationShard needs useDeviceUpdate and lambda a multiple of world");
    cfg.lambda_total = d.lambda;
    cfg.lambda_local = d.lambda / d.world;
    cfg.first_candidate = d.rank * cfg.lambda_local;
    cfg.m = d.m;
    cfg.mu = d.mu;
    cfg.z_source = d.philox ? LMCMA_Z_PHILOX : LMCMA_Z_HOST;
    cfg.cost_source = d.haveProblem ? LMCMA_COST_TRAJECTORY : LMCMA_COST_EXTERNAL;
    cfg.seed = (uint64_t)(d.inseed < 1 ? d.rng.startseed : d.inseed);
    if (d.haveProblem) {
        cfg.pad_mode = d.problem.pad_mode;
        cfg.w_len = d.problem.w_len;
        cfg.w_grid = d.problem.w_grid;
        cfg.w_smooth = d.problem.w_smooth;
        for (int k = 0; k < 3; ++k) {
            cfg.start[k] = d.problem.start[k];
            cfg.goal[k] = d.problem.goal[k];
        }
    }
    d.hip(lmcma_hip_create(&cfg, &d.ctx), "lmcma_hip_create");
    if (d.loBounds || d.hiBounds) d.hip(lmcma_hip_set_bounds(d.ctx, d.loBounds, d.hiBounds), "lmcma_hip_set_bounds");
    // normals drawn on the device get the factor there; host-drawn ones in sample()
    if (d.philox && !d.Lchol.empty()) d.hip(lmcma_hip_set_cov_factor(d.ctx, d.Lchol.data()), "lmcma_hip_set_cov_factor");
    if (d.haveProblem)
        d.hip(lmcma_hip_set_grid(d.ctx, d.problem.cells, d.problem.nx, d.problem.ny, d.problem.nz,
                                 d.problem.cell_type, d.problem.c_obs),
              "lmcma_hip_set_grid");
    d.initialised = true;
    if (sharded) d.hip(lmcma_hip_comm_init(d.ctx, d.uniqueId.data(), d.world, d.rank), "lmcma_hip_comm_init");
    if (d.deviceUpdate) {
        d.hip(lmcma_hip_opt_init(d.ctx, d.xmean.data(), d.sigma), "lmcma_hip_opt_init");
    } else {
        sample();  // the first population is drawn inside init (L/:298)
    }
}

void LMCMA::sample()
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("LMCMA::sample before init");
    if (d.deviceUpdate) throw std::runtime_error("LMCMA::sample: the device update samples inside step()");
    const int N = d.N, k = d.active;
    // active direction pairs in iterator order (what computeAz walks, L/:435-439)
    std::vector<double> V((size_t)std::max(k, 1) * N), P((size_t)std::max(k, 1) * N), coef(std::max(k, 1));
    for (int t = 0; t < k; ++t) {
        const int sl = d.order[t];
        std::copy(&d.vArr[(size_t)sl * N], &d.vArr[(size_t)sl * N] + N, &V[(size_t)t * N]);
        std::copy(&d.pcArr[(size_t)sl * N], &d.pcArr[(size_t)sl * N] + N, &P[(size_t)t * N]);
        coef[t] = d.nj[sl];
    }
    d.hip(lmcma_hip_set_state(d.ctx, d.xmean.data(), d.sigma, V.data(), P.data(), coef.data(), k, d.M),
          "lmcma_hip_set_state");
    if (d.philox) {
        d.hip(lmcma_hip_sample(d.ctx, (uint32_t)d.itr, 0), "lmcma_hip_sample");
    } else {
        // candidate by candidate, N normals each, then the optional factor (L/:212-218)
        std::vector<double> Z((size_t)d.lambda * N);
        for (int i = 0; i < d.lambda; ++i) {
            double *z = &Z[(size_t)i * N];
            for (int n = 0; n < N; ++n) z[n] = random_Gauss(&d.rng);
            if (!d.Lchol.empty()) applyCovL(d.Lchol.data(), z, N);
        }
        d.hip(lmcma_hip_sample(d.ctx, (uint32_t)d.itr, Z.data()), "lmcma_hip_sample");
    }
    d.rowsFresh = false;
}

void LMCMA::askHook(double *params, int N)
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("LMCMA::getNextParameterVector before init");
    if (d.deviceUpdate && !d.popReady) {
        // the state lives on the device: draw this generation's candidates there (the Philox
        // counter is the generation number, as inside lmcma_hip_run)
        if (!d.uniqueId.empty()) throw std::runtime_error("ask/tell is not available for a shard of a population; use step()");
        d.hip(lmcma_hip_sample(d.ctx, (uint32_t)d.itr, 0), "lmcma_hip_sample");
        d.rowsFresh = false;
        d.popReady = true;
    }
    d.fetchRows();
    const int n = std::min(N, d.N);
    std::copy(&d.rows[(size_t)d.sampleIdx * d.N], &d.rows[(size_t)d.sampleIdx * d.N] + n, params);
}

void LMCMA::tellHook(double *feedbacks, int numFeedbacks)
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("LMCMA::setEvaluationFeedback before init");
    if (d.deviceUpdate && !d.popReady) throw std::runtime_error("setEvaluationFeedback before getNextParameterVector");
    double f = 0.0;
    for (int i = 0; i < numFeedbacks; ++i) f += feedbacks[i];  // left to right, L/:186-188
    d.fitness[d.sampleIdx] = f;
    d.lastCosts[d.sampleIdx] = f;
    ++counteval;
    if (f < BestF || counteval == 1) {
        BestF = f;
        if (d.verbose) std::cout << "Functions evaluation #" << counteval << ", value: " << f << std::endl;
    }
    if (++d.sampleIdx % d.lambda == 0) {
        if (d.deviceUpdate) {
            // the caller's costs go to the device (ranked there as floats), which ranks, recombines
            // and updates; the next generation is drawn at the next getNextParameterVector
            lmcma_hip_opt_summary sm;
            d.hip(lmcma_hip_set_costs(d.ctx, 0, d.lambda, d.fitness.data()), "lmcma_hip_set_costs");
            d.hip(lmcma_hip_phase_rank(d.ctx), "lmcma_hip_phase_rank");
            d.hip(lmcma_hip_phase_update(d.ctx, 0), "lmcma_hip_phase_update");
            d.hip(lmcma_hip_opt_get(d.ctx, &sm, d.xmean.data()), "lmcma_hip_opt_get");
            d.itr = sm.itr;
            d.sigma = sm.sigma;
            d.s = sm.s;
            d.active = sm.iterator_sz;
            d.popReady = false;
        } else {
            update();
            sample();
        }
        d.sampleIdx = 0;
    }
}

namespace {

// The state transition shared by both feedback paths: everything of L/:327-423
// after the new mean is known.  sorted = this generation's costs ascending.
void advanceState(LMCMA::Impl &d, const std::vector<double> &sorted)
{
    const int N = d.N, m = d.m, lambda = d.lambda;
    // evolution path, L/:327-329
    const double gain = sqrt(d.cc * (2 - d.cc) * d.mueff);
    for (int i = 0; i < N; ++i)
        d.pc[i] = (1 - d.cc) * d.pc[i] + gain * (d.xmean[i] - d.xold[i]) / d.sigma;

    // which stored direction makes room, L/:331-356: while the store fills, append;
    // afterwards drop the younger of the closest-in-age neighbours (or the oldest
    // when every gap is already >= m) and rotate it to the back.
    int first = 1;
    if (d.itr < m) {
        d.slot[d.itr] = d.itr;
    } else if (m > 1) {
        int gap = d.born[d.slot[1]] - d.born[d.slot[0]];
        for (int j = 1; j < m - 1; ++j) {
            const int g = d.born[d.slot[j + 1]] - d.born[d.slot[j]];
            if (g < gap) {
                gap = g;
                first = j + 1;
            }
        }
        if (gap >= m) first = 0;
        if (first != m - 1) std::rotate(d.slot.begin() + first, d.slot.begin() + first + 1, d.slot.end());
    } else {
        first = 0;
    }
    d.active = std::min(d.itr + 1, m);
    std::copy(d.slot.begin(), d.slot.begin() + d.active, d.order.begin());
    const int fresh = d.slot[d.active - 1];
    d.born[fresh] = d.itr;
    std::copy(d.pc.begin(), d.pc.end(), d.pcArr.begin() + (size_t)fresh * N);

    // direction vectors from `first` on are rebuilt: v_i = T_{i-1} ... T_0 pc_i with
    // T_j = K I - lj_j v_j v_j^T (invAz, L/:449-463), then their coefficients (L/:383-389)
    if (first == 1) first = 0;  // L/:373-374
    std::vector<double> av(N);
    for (int i = first; i < d.active; ++i) {
        const int sl = d.slot[i];
        std::copy(d.pcArr.begin() + (size_t)sl * N, d.pcArr.begin() + (size_t)(sl + 1) * N, av.begin());
        for (int j = 0; j < i; ++j) {
            const double *vj = &d.vArr[(size_t)d.order[j] * N];
            double dot = 0;
            for (int p = 0; p < N; ++p) dot += vj[p] * av[p];
            dot = d.lj[d.order[j]] * dot;
            for (int p = 0; p < N; ++p) av[p] = d.K * av[p] - dot * vj[p];
        }
        std::copy(av.begin(), av.end(), d.vArr.begin() + (size_t)sl * N);
        double nv = 0;
        for (int p = 0; p < N; ++p) nv += av[p] * av[p];
        d.nj[sl] = (sqrt(1 - d.c1) / nv) * (sqrt(1 + (d.c1 / (1 - d.c1)) * nv) - 1);
        d.lj[sl] = (1 / (sqrt(1 - d.c1) * nv)) * (1 - (1 / sqrt(1 + (d.c1 / (1 - d.c1)) * nv)));
    }

    // population success rule, L/:393-419.  The reference stable-sorts
    // [current | previous] and averages the ranks of each half; its variable
    // "meanprev" holds the current half's mean rank and "meancur" the previous
    // half's, so the step size grows when the current generation ranks better.
    if (d.itr > 0) {
        std::vector<double> mixed(2 * (size_t)lambda);
        std::vector<int> idx(2 * (size_t)lambda);
        std::vector<sortedvals> tmp(2 * (size_t)lambda);
        std::copy(sorted.begin(), sorted.end(), mixed.begin());
        std::copy(d.prevSorted.begin(), d.prevSorted.end(), mixed.begin() + lambda);
        myqsort(2 * lambda, mixed.data(), idx.data(), tmp.data());
        double rankCurrent = 0, rankPrevious = 0;
        std::vector<int> rank(2 * (size_t)lambda);
        for (int r = 0; r < 2 * lambda; ++r) rank[idx[r]] = r;
        for (int i = 0; i < lambda; ++i) {
            rankCurrent = rankCurrent + rank[i];
            rankPrevious = rankPrevious + rank[lambda + i];
        }
        rankCurrent = rankCurrent / lambda;
        rankPrevious = rankPrevious / lambda;
        const double success = (rankPrevious - rankCurrent) / lambda;
        d.s = (1 - d.cs) * d.s + d.cs * (success - d.target);
        d.sigma = d.sigma * exp(d.s / 1.0);
    }
    d.prevSorted = sorted;
    d.itr++;
}

}  // namespace

void LMCMA::update()
{
    // feedback path of ask/tell: costs are the caller's doubles, so the ranking
    // (L/:315) and the recombination (L/:316-326) run on the host in fp64.
    Impl &d = *impl;
    const int N = d.N;
    std::vector<int> idx(d.lambda);
    std::vector<sortedvals> tmp(d.lambda);
    std::vector<double> sorted(d.fitness);
    myqsort(d.lambda, sorted.data(), idx.data(), tmp.data());
    d.fetchRows();
    d.xold = d.xmean;
    std::fill(d.xmean.begin(), d.xmean.end(), 0.0);
    for (int i = 0; i < d.mu; ++i) {
        const double *x = &d.rows[(size_t)idx[i] * N];
        for (int j = 0; j < N; ++j) d.xmean[j] += d.weights[i] * x[j];
    }
    advanceState(d, sorted);
}

void LMCMA::evaluatePopulation() { step(1); }

void LMCMA::step(int generations)
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("LMCMA::step before init");
    if (!d.haveProblem) throw std::runtime_error("step needs setTrajectoryProblem");
    if (generations < 1) return;
    if (d.sampleIdx != 0) throw std::runtime_error("step in the middle of an ask/tell generation");
    if (d.deviceUpdate) {
        // the library replays captured generations back to back; the stop rule (L/:426-429) is
        // evaluated on the device and freezes the state, and it is looked at here every 512
        // generations so that a run that has stopped returns early
        lmcma_hip_opt_summary sm;
        for (int left = generations; left > 0;) {
            const int n = left < 512 ? left : 512;
            if (d.uniqueId.empty()) d.hip(lmcma_hip_run(d.ctx, n), "lmcma_hip_run");
            else d.hip(lmcma_hip_run_sharded(d.ctx, n), "lmcma_hip_run_sharded");
            left -= n;
            d.hip(lmcma_hip_opt_get(d.ctx, &sm, d.xmean.data()), "lmcma_hip_opt_get");
            if (sm.done) break;
        }
        d.itr = sm.itr;
        d.sigma = sm.sigma;
        d.s = sm.s;
        d.active = sm.iterator_sz;
        counteval = (int)sm.counteval;
        BestF = sm.best_f;
        d.rowsFresh = false;
        d.popReady = false;
        return;
    }
    for (int g = 0; g < generations; ++g) {
        if (d.sampleIdx != 0) throw std::runtime_error("evaluatePopulation in the middle of an ask/tell generation");
        // device: score, rank, recombine; host: the fp64 state transition
        d.hip(lmcma_hip_evaluate(d.ctx), "lmcma_hip_evaluate");
        std::vector<float> sortedF(d.lambda);
        std::vector<int32_t> perm(d.lambda);
        d.hip(lmcma_hip_select(d.ctx, sortedF.data(), perm.data()), "lmcma_hip_select");
        d.xold = d.xmean;
        d.hip(lmcma_hip_recombine(d.ctx, d.weights.data(), d.xmean.data()), "lmcma_hip_recombine");
        std::vector<double> sorted(sortedF.begin(), sortedF.end());
        // bookkeeping of lambda feedback calls (L/:190-198); the smallest cost decides BestF
        for (int i = 0; i < d.lambda; ++i) d.lastCosts[perm[i]] = sorted[i];
        const bool firstEver = counteval == 0;
        counteval += d.lambda;
        if (sorted[0] < BestF || firstEver) {
            BestF = sorted[0];
            if (d.verbose) std::cout << "Functions evaluation #" << counteval << ", value: " << BestF << std::endl;
        }
        advanceState(d, sorted);
        sample();
    }
}

bool LMCMA::isBehaviorLearningDone() { return impl->sigma < 1e-20; }  // L/:426-429

int LMCMA::dimension() const { return impl->N; }
int LMCMA::populationSize() const { return impl->lambda; }
int LMCMA::directionVectors() const { return impl->m; }
int LMCMA::generation() const { return impl->itr; }
double LMCMA::stepSize() const { return impl->sigma; }
void LMCMA::mean(double *out) const { std::copy(impl->xmean.begin(), impl->xmean.end(), out); }
void LMCMA::population(double *out) const
{
    impl->fetchRows();
    std::copy(impl->rows.begin(), impl->rows.end(), out);
}
void LMCMA::lastCosts(double *out) const { std::copy(impl->lastCosts.begin(), impl->lastCosts.end(), out); }
const char *LMCMA::lastError() const { return impl->error.c_str(); }
