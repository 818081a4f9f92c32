// Separable CMA-ES behind the reference's class interface (lmcma_path_planner/src/lmcma.hpp:152-190);
// L/ below = lmcma_path_planner/src/lmcma.cpp.  The population is drawn (and, with a trajectory
// problem, scored) on the device through the C ABI of lmcma_hip.h; everything of
// SepCMA::update (L/:530-577) runs here in fp64 in the reference's order of operations.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmcma.hpp"
#include "lmcma_hip.h"

struct SepCMA::Impl {
    double *initialParams, *loBounds, *hiBounds, *covarianceIn;
    int lambdaArg, inseed, device = 0;
    bool verbose, haveProblem = false;
    LMCMATrajectoryProblem problem;
    std::vector<uint8_t> cellStore;

    int N = 0, lambda = 0, mu = 0, itr = 0, sampleIdx = 0;
    double mueff = 0, sigma = 1.0, cc = 0, c1 = 0, cmu = 0, ccov1 = 0, ccovmu = 0, chiN = 0, cs = 0, damps = 0;
    std::vector<double> weights, xmean, xold, arz, diagD, diagC, pc, ps, Lchol, fitness, rows;
    bool rowsFresh = false;
    random_t rng;
    lmcma_hip_ctx *ctx = 0;
    bool initialised = false;

    void hip(int rc, const char *what)
    {
        if (rc != LMCMA_HIP_OK) throw std::runtime_error(std::string(what) + ": " + lmcma_hip_last_error());
    }
    void fetchRows()
    {
        if (rowsFresh) return;
        hip(lmcma_hip_get_rows(ctx, 0, lambda, rows.data()), "lmcma_hip_get_rows");
        rowsFresh = true;
    }
};

SepCMA::SepCMA(double *initialParams, int lambda, double *loBounds, double *hiBounds, double sigma,
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
    random_init(&impl->rng, (long unsigned)(inseed < 1 ? 0 : inseed));  // seeded at construction (L/:114)
}

SepCMA::~SepCMA()
{
    if (impl->ctx) lmcma_hip_destroy(impl->ctx);
    random_exit(&impl->rng);
    delete impl;
}

void SepCMA::setDevice(int ordinal) { impl->device = ordinal; }

void SepCMA::setTrajectoryProblem(const LMCMATrajectoryProblem &p)
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

void SepCMA::init(int N)
{
    Impl &d = *impl;
    if (d.initialised) throw std::runtime_error("SepCMA::init called twice");
    if (N < 1) throw std::invalid_argument("SepCMA::init: N must be positive");
    d.N = N;
    // CMABase::init, L/:134-169
    d.lambda = d.lambdaArg < 1 ? 4 + (int)(3 * log((double)N)) : d.lambdaArg;
    d.mu = d.lambda / 2;
    if (d.mu < 1) throw std::invalid_argument("SepCMA::init: lambda must be at least 2");
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
    else for (int i = 0; i < N; ++i) d.xmean[i] = random_Uniform(&d.rng);
    if (d.covarianceIn) {
        d.Lchol.resize((size_t)N * N);
        cholesky(d.covarianceIn, d.Lchol.data(), N);
    }
    // SepCMA::init, L/:485-517
    d.arz.assign((size_t)N * d.lambda, 0.0);
    d.diagD.assign(N, 1.0);
    d.diagC.assign(N, 1.0);
    d.pc.assign(N, 0.0);
    d.ps.assign(N, 0.0);
    d.fitness.assign(d.lambda, 0.0);
    d.rows.assign((size_t)d.lambda * N, 0.0);
    d.cc = 1.0 / d.lambda;
    d.c1 = 2.0 / (pow(N + 1.3, 2.0) + d.mueff);
    d.cmu = std::min(1.0 - d.c1, 2.0 * (d.mueff - 2.0 + 1.0 / d.mueff) / (pow(N + 2.0, 2.0) + d.mueff));
    d.ccov1 = std::min(1.0, d.c1 * (N + 1.5) / 3.0);
    d.ccovmu = std::min(1.0 - d.ccov1, d.cmu * (N + 1.5) / 3);
    d.chiN = sqrt((double)N) * (1.0 - 1.0 / (4.0 * N) + 1 / (21.0 * N * N));
    d.cs = (d.mueff + 2.0) / (N + d.mueff + 5.0);
    d.damps = 1.0 + 2 * std::max(0.0, sqrt((d.mueff - 1.0) / (N + 1.0)) - 1.0) + d.cs;

    if (d.haveProblem && d.problem.dims * d.problem.waypoints != N)
        throw std::invalid_argument("SepCMA::init: N must equal dims * waypoints of the trajectory problem");
    lmcma_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = d.device;
    cfg.dims = d.haveProblem ? d.problem.dims : 0;
    cfg.waypoints = d.haveProblem ? d.problem.waypoints : 0;
    cfg.n_params = N;
    cfg.lambda_total = cfg.lambda_local = d.lambda;
    cfg.m = 1;  // no direction store: the covariance is the diagonal folded into z
    cfg.mu = d.mu;
    cfg.z_source = LMCMA_Z_HOST;
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
    if (d.haveProblem)
        d.hip(lmcma_hip_set_grid(d.ctx, d.problem.cells, d.problem.nx, d.problem.ny, d.problem.nz,
                                 d.problem.cell_type, d.problem.c_obs),
              "lmcma_hip_set_grid");
    d.initialised = true;
    sample();  // L/:516
}

void SepCMA::sample()
{
    // L/:519-528: arx = xmean + sigma D z, then the bounds.  The device computes
    // xmean + sigma z' for the normals it is handed, so D is folded into z' = D z here.
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("SepCMA::sample before init");
    const int N = d.N;
    std::vector<double> scaled((size_t)d.lambda * N);
    for (int i = 0; i < d.lambda; ++i) {
        double *z = &d.arz[(size_t)i * N];
        for (int n = 0; n < N; ++n) z[n] = random_Gauss(&d.rng);
        if (!d.Lchol.empty()) applyCovL(d.Lchol.data(), z, N);
        for (int n = 0; n < N; ++n) scaled[(size_t)i * N + n] = d.diagD[n] * z[n];
    }
    d.hip(lmcma_hip_set_state(d.ctx, d.xmean.data(), d.sigma, 0, 0, 0, 0, 1.0), "lmcma_hip_set_state");
    d.hip(lmcma_hip_sample(d.ctx, (uint32_t)d.itr, scaled.data()), "lmcma_hip_sample");
    d.rowsFresh = false;
}

void SepCMA::askHook(double *params, int N)
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("SepCMA::getNextParameterVector before init");
    d.fetchRows();
    const int n = std::min(N, d.N);
    std::copy(&d.rows[(size_t)d.sampleIdx * d.N], &d.rows[(size_t)d.sampleIdx * d.N] + n, params);
}

void SepCMA::tellHook(double *feedbacks, int numFeedbacks)
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("SepCMA::setEvaluationFeedback before init");
    double f = 0.0;
    for (int i = 0; i < numFeedbacks; ++i) f += feedbacks[i];  // L/:186-188
    d.fitness[d.sampleIdx] = f;
    ++counteval;
    if (f < BestF || counteval == 1) {
        BestF = f;
        if (d.verbose) std::cout << "Functions evaluation #" << counteval << ", value: " << f << std::endl;
    }
    if (++d.sampleIdx % d.lambda == 0) {
        update();
        sample();
        d.sampleIdx = 0;
    }
}

void SepCMA::update()
{
    // L/:530-577
    Impl &d = *impl;
    const int N = d.N, mu = d.mu;
    std::vector<int> idx(d.lambda);
    std::vector<sortedvals> tmp(d.lambda);
    std::vector<double> sorted(d.fitness);
    myqsort(d.lambda, sorted.data(), idx.data(), tmp.data());
    d.fetchRows();
    for (int i = 0; i < N; ++i) {
        d.xold[i] = d.xmean[i];
        d.xmean[i] = 0;
    }
    for (int i = 0; i < mu; ++i) {
        const double *x = &d.rows[(size_t)idx[i] * N];
        for (int j = 0; j < N; ++j) d.xmean[j] += d.weights[i] * x[j];
    }
    double norm_ps = 0;
    for (int i = 0; i < N; ++i) {
        d.ps[i] = (1 - d.cs) * d.ps[i] +
                  sqrt(d.cs * (2 - d.cs) * d.mueff) * (1. / d.diagD[i]) * (d.xmean[i] - d.xold[i]) / d.sigma;
        norm_ps += d.ps[i] * d.ps[i];
    }
    norm_ps = sqrt(norm_ps);
    for (int i = 0; i < N; ++i)
        d.pc[i] = (1 - d.cc) * d.pc[i] + sqrt(d.cc * (2 - d.cc) * d.mueff) * (d.xmean[i] - d.xold[i]) / d.sigma;
    for (int i = 0; i < N; ++i) {
        double val = 0;
        for (int j = 0; j < mu; ++j) {
            const double z = d.arz[(size_t)idx[j] * N + i];
            val += d.weights[j] * z * z;
        }
        d.diagC[i] = (1 - d.ccov1 - d.ccovmu) * d.diagC[i] + d.ccov1 * d.pc[i] * d.pc[i] +
                     d.ccovmu * (d.diagC[i] * val);
    }
    d.sigma *= exp((d.cs / d.damps) * (norm_ps / d.chiN - 1));
    for (int i = 0; i < N; ++i) d.diagD[i] = sqrt(d.diagC[i]);
    d.itr++;
}

void SepCMA::evaluatePopulation()
{
    Impl &d = *impl;
    if (!d.initialised) throw std::runtime_error("SepCMA::evaluatePopulation before init");
    if (!d.haveProblem) throw std::runtime_error("evaluatePopulation needs setTrajectoryProblem");
    if (d.sampleIdx != 0) throw std::runtime_error("evaluatePopulation in the middle of an ask/tell generation");
    d.hip(lmcma_hip_evaluate(d.ctx), "lmcma_hip_evaluate");
    d.hip(lmcma_hip_get_costs(d.ctx, 0, d.lambda, d.fitness.data()), "lmcma_hip_get_costs");
    // the bookkeeping of lambda feedback calls, in candidate order (L/:190-198)
    for (int i = 0; i < d.lambda; ++i) {
        ++counteval;
        if (d.fitness[i] < BestF || counteval == 1) BestF = d.fitness[i];
    }
    update();
    sample();
}

int SepCMA::dimension() const { return impl->N; }
int SepCMA::populationSize() const { return impl->lambda; }
int SepCMA::generation() const { return impl->itr; }
double SepCMA::stepSize() const { return impl->sigma; }
void SepCMA::mean(double *out) const { std::copy(impl->xmean.begin(), impl->xmean.end(), out); }
void SepCMA::diagonal(double *out) const { std::copy(impl->diagD.begin(), impl->diagD.end(), out); }
void SepCMA::population(double *out) const
{
    impl->fetchRows();
    std::copy(impl->rows.begin(), impl->rows.end(), out);
}
