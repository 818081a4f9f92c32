// C bindings of the LMCMA class and of the host helpers (include/lmcma_opt.h).
#include "lmcma_opt.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmcma.hpp"
#include "lmcma_hip.h"

namespace {

thread_local std::string g_err;

template <typename F>
int guarded(F &&f)
{
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
    } catch (...) {
        g_err = "unknown failure";
    }
    return -1;
}

}  // namespace

struct lmcma_opt {
    std::vector<double> initial, lo, hi, cov;
    int n = 0;               // entries of initial / lo / hi (cov: n * n): the only N init() may be given
    LMCMA *opt = nullptr;
    SepCMA *sep = nullptr;   // a handle holds one of the three optimisers
    CMAChol *chol = nullptr;
    ~lmcma_opt() { delete opt; delete sep; delete chol; }
    CMABase *base()
    {
        if (opt) return opt;
        if (sep) return sep;
        return chol;
    }
};

extern "C" {

const char *lmcma_opt_last_error(void) { return g_err.c_str(); }

int lmcma_opt_create(const double *initial, const double *lo, const double *hi, const double *cov,
                     int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out)
{
    if (!out || n < 1) { g_err = "lmcma_opt_create: bad arguments"; return -1; }
    return guarded([&] {
        lmcma_opt *h = new lmcma_opt();
        h->n = n;
        if (initial) h->initial.assign(initial, initial + n);
        if (lo) h->lo.assign(lo, lo + n);
        if (hi) h->hi.assign(hi, hi + n);
        if (cov) h->cov.assign(cov, cov + (size_t)n * n);
        h->opt = new LMCMA(initial ? h->initial.data() : 0, lambda, lo ? h->lo.data() : 0,
                           hi ? h->hi.data() : 0, sigma, cov ? h->cov.data() : 0, seed, verbose != 0);
        *out = h;
    });
}

int lmcma_opt_create_sep(const double *initial, const double *lo, const double *hi, const double *cov,
                         int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out)
{
    if (!out || n < 1) { g_err = "lmcma_opt_create_sep: bad arguments"; return -1; }
    return guarded([&] {
        lmcma_opt *h = new lmcma_opt();
        h->n = n;
        if (initial) h->initial.assign(initial, initial + n);
        if (lo) h->lo.assign(lo, lo + n);
        if (hi) h->hi.assign(hi, hi + n);
        if (cov) h->cov.assign(cov, cov + (size_t)n * n);
        h->sep = new SepCMA(initial ? h->initial.data() : 0, lambda, lo ? h->lo.data() : 0,
                            hi ? h->hi.data() : 0, sigma, cov ? h->cov.data() : 0, seed, verbose != 0);
        *out = h;
    });
}

int lmcma_opt_create_chol(const double *initial, const double *lo, const double *hi, const double *cov,
                          int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out)
{
    if (!out || n < 1) { g_err = "lmcma_opt_create_chol: bad arguments"; return -1; }
    return guarded([&] {
        lmcma_opt *h = new lmcma_opt();
        h->n = n;
        if (initial) h->initial.assign(initial, initial + n);
        if (lo) h->lo.assign(lo, lo + n);
        if (hi) h->hi.assign(hi, hi + n);
        if (cov) h->cov.assign(cov, cov + (size_t)n * n);
        h->chol = new CMAChol(initial ? h->initial.data() : 0, lambda, lo ? h->lo.data() : 0,
                              hi ? h->hi.data() : 0, sigma, cov ? h->cov.data() : 0, seed, verbose != 0);
        *out = h;
    });
}

int lmcma_opt_destroy(lmcma_opt *o)
{
    return guarded([&] { delete o; });
}

int lmcma_opt_set_options(lmcma_opt *o, int m, int device, int philox, int device_update)
{
    if (!o) { g_err = "null handle"; return -1; }
    return guarded([&] {
        if (!o->opt) {
            if (philox || device_update) throw std::invalid_argument("SepCMA / CMAChol: host normals and host update only");
            if (o->sep) o->sep->setDevice(device);
            else o->chol->setDevice(device);
            return;
        }
        o->opt->setDirectionVectors(m);
        o->opt->setDevice(device);
        o->opt->usePhiloxSampling(philox != 0);
        o->opt->useDeviceUpdate(device_update != 0);
    });
}

int lmcma_opt_set_shard(lmcma_opt *o, int world, int rank, const char *unique_id)
{
    if (!o) { g_err = "null handle"; return -1; }
    return guarded([&] {
        if (!o->opt) throw std::invalid_argument("SepCMA / CMAChol: one process only");
        o->opt->setPopulationShard(world, rank, unique_id);
    });
}

int lmcma_opt_set_problem(lmcma_opt *o, int dims, int waypoints, int nx, int ny, int nz, int cell_type,
                          const void *cells, double c_obs, const double *start, const double *goal,
                          double w_len, double w_grid, double w_smooth, int pad_mode)
{
    if (!o || !start || !goal) { g_err = "null argument"; return -1; }
    return guarded([&] {
        LMCMATrajectoryProblem p;
        p.dims = dims;
        p.waypoints = waypoints;
        p.nx = nx; p.ny = ny; p.nz = nz;
        p.cell_type = cell_type;
        p.cells = cells;
        p.c_obs = c_obs;
        for (int d = 0; d < 3; ++d) {
            p.start[d] = d < dims ? start[d] : 0.0;
            p.goal[d] = d < dims ? goal[d] : 0.0;
        }
        p.w_len = w_len; p.w_grid = w_grid; p.w_smooth = w_smooth;
        p.pad_mode = pad_mode;
        if (o->sep) o->sep->setTrajectoryProblem(p);
        else if (o->chol) o->chol->setTrajectoryProblem(p);
        else o->opt->setTrajectoryProblem(p);
    });
}

int lmcma_opt_init(lmcma_opt *o, int N)
{
    if (!o) { g_err = "null handle"; return -1; }
    // init reads N entries of the arrays the handle copied at creation
    if (N != o->n) { g_err = "lmcma_opt_init: N differs from the n the handle was created with"; return -1; }
    return guarded([&] { o->base()->init(N); });
}

int lmcma_opt_ask(lmcma_opt *o, double *params, int N)
{
    if (!o || !params) { g_err = "null argument"; return -1; }
    return guarded([&] { o->base()->getNextParameterVector(params, N); });
}

int lmcma_opt_tell(lmcma_opt *o, const double *feedbacks, int num_feedbacks)
{
    if (!o || !feedbacks) { g_err = "null argument"; return -1; }
    return guarded([&] { o->base()->setEvaluationFeedback(const_cast<double *>(feedbacks), num_feedbacks); });
}

int lmcma_opt_done(lmcma_opt *o)
{
    if (!o) { g_err = "null handle"; return -1; }
    int done = 0;
    int rc = guarded([&] { done = o->base()->isBehaviorLearningDone() ? 1 : 0; });
    return rc < 0 ? rc : done;
}

int lmcma_opt_step(lmcma_opt *o, int generations)
{
    if (!o) { g_err = "null handle"; return -1; }
    return guarded([&] {
        if (o->sep) o->sep->step(generations);
        else if (o->chol) o->chol->step(generations);
        else o->opt->step(generations);
    });
}

int lmcma_opt_get_info(lmcma_opt *o, lmcma_opt_info *info)
{
    if (!o || !info) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (o->sep) {
            info->n = o->sep->dimension();
            info->lambda = o->sep->populationSize();
            info->m = 0;
            info->generation = o->sep->generation();
            info->counteval = o->sep->counteval;
            info->best_f = o->sep->BestF;
            info->sigma = o->sep->stepSize();
            return;
        }
        if (o->chol) {
            info->n = o->chol->dimension();
            info->lambda = o->chol->populationSize();
            info->m = 0;
            info->generation = o->chol->generation();
            info->counteval = o->chol->counteval;
            info->best_f = o->chol->BestF;
            info->sigma = o->chol->stepSize();
            return;
        }
        info->n = o->opt->dimension();
        info->lambda = o->opt->populationSize();
        info->m = o->opt->directionVectors();
        info->generation = o->opt->generation();
        info->counteval = o->opt->counteval;
        info->best_f = o->opt->BestF;
        info->sigma = o->opt->stepSize();
    });
}

int lmcma_opt_get_mean(lmcma_opt *o, double *xmean)
{
    if (!o || !xmean) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (o->sep) o->sep->mean(xmean);
        else if (o->chol) o->chol->mean(xmean);
        else o->opt->mean(xmean);
    });
}

int lmcma_opt_get_population(lmcma_opt *o, double *rows)
{
    if (!o || !rows) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (o->sep) o->sep->population(rows);
        else if (o->chol) o->chol->population(rows);
        else o->opt->population(rows);
    });
}

int lmcma_opt_get_last_costs(lmcma_opt *o, double *costs)
{
    if (!o || !costs) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (!o->opt) throw std::invalid_argument("last costs are kept by LMCMA handles only");
        o->opt->lastCosts(costs);
    });
}

int lmcma_opt_get_factor(lmcma_opt *o, double *a_rowmajor)
{
    if (!o || !a_rowmajor) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (!o->chol) throw std::invalid_argument("only CMAChol handles have a factor");
        o->chol->factor(a_rowmajor);
    });
}

int lmcma_opt_get_diagonal(lmcma_opt *o, double *diag_d)
{
    if (!o || !diag_d) { g_err = "null argument"; return -1; }
    return guarded([&] {
        if (!o->sep) throw std::invalid_argument("only SepCMA handles have a diagonal");
        o->sep->diagonal(diag_d);
    });
}

/* ---- host helpers ------------------------------------------------------------------- */

int lmcma_host_gauss_stream(int seed, int n, double *out)
{
    if (seed < 1 || n < 0 || !out) { g_err = "seed must be >= 1"; return -1; }
    random_t r;
    random_init(&r, (long unsigned)seed);
    for (int i = 0; i < n; ++i) out[i] = random_Gauss(&r);
    random_exit(&r);
    return 0;
}

int lmcma_host_uniform_stream(int seed, int n, double *out)
{
    if (seed < 1 || n < 0 || !out) { g_err = "seed must be >= 1"; return -1; }
    random_t r;
    random_init(&r, (long unsigned)seed);
    for (int i = 0; i < n; ++i) out[i] = random_Uniform(&r);
    random_exit(&r);
    return 0;
}

int lmcma_host_covariance(int num_dims, int num_waypoints, double *cov)
{
    return guarded([&] { covariance(num_dims, num_waypoints, cov); });
}

int lmcma_host_differentiation_matrix(int num_time_steps, int order, double dt, double *diff_matrix, int row_len)
{
    if (order < 0 || order > 3) { g_err = "order must be 0..3"; return -1; }
    return guarded([&] { differentiationMatrix(num_time_steps, order, dt, diff_matrix, row_len); });
}

int lmcma_host_cholesky(const double *C, double *L, int N)
{
    return guarded([&] { cholesky(const_cast<double *>(C), L, N); });
}

int lmcma_host_apply_cov_l(const double *L, double *z, int N)
{
    return guarded([&] { applyCovL(const_cast<double *>(L), z, N); });
}

int lmcma_host_sort(int n, double *values, int32_t *index)
{
    return guarded([&] {
        std::vector<sortedvals> tmp(n);
        std::vector<int> idx(n);
        myqsort(n, values, idx.data(), tmp.data());
        std::copy(idx.begin(), idx.end(), index);
    });
}

}  // extern "C"
