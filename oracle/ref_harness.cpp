// Harness around the UNMODIFIED reference optimiser (lmcma_path_planner/src/lmcma.cpp of the
// reference tree), compiled where it lies by `make -C oracle ref` into oracle/_ref/.
//
// TEST INFRASTRUCTURE ONLY.  It exists to (1) write the known-answer fixtures under tests/golden/
// that pin oracle/lmcma_oracle.c, (2) let the CPU tests compare the oracle with the reference live
// when the library is present, (3) serve as bench.py's "reference" CPU baseline leg.  Nothing under
// lmcma_path_planner_amd/ loads it.
//
// What is compiled of the reference: lines 1-5 and 7-835 of lmcma.cpp, i.e. every line except
// `#include <eigen3/Eigen/Dense>` (:6) and the three functions that use Eigen (invert, cholesky,
// applyCovL, :836-864).  Eigen is not installed here and nothing stands in for it: the three symbols
// stay without code (the link line binds them to address 0), so this harness never passes a
// covariance matrix (the only way the optimisers reach them, lmcma.cpp:165-169, 216-217) and never
// calls covariance() (:769-810, which calls invert).
//
// This file contains no reference code: it includes the reference's own header through -I and reads
// the optimisers' private state through the access-specifier macros below.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

// members that follow `class X {` without a keyword are private by default: hence `class` too
// (struct and class mangle alike, so the objects link against the reference's translation unit)
#define class struct
#define private public
#define protected public
#include "lmcma.hpp"  // the reference's header (lmcma_path_planner/src/lmcma.hpp)
#undef private
#undef protected
#undef class

namespace {

struct Args {
    std::vector<double> initial, lo, hi;
    double* initial_p = nullptr;
    double* lo_p = nullptr;
    double* hi_p = nullptr;
    Args(int N, const double* initial_, const double* lo_, const double* hi_)
    {
        if (initial_) { initial.assign(initial_, initial_ + N); initial_p = initial.data(); }
        if (lo_) { lo.assign(lo_, lo_ + N); lo_p = lo.data(); }
        if (hi_) { hi.assign(hi_, hi_ + N); hi_p = hi.data(); }
    }
};

template <typename Opt>
struct Handle {
    Args args;  // the reference keeps the caller's pointers (lmcma.cpp:109-110): they live here
    Opt opt;
    int N;
    Handle(int N_, int lambda, const double* initial, const double* lo, const double* hi, double sigma, int seed)
        : args(N_, initial, lo, hi), opt(args.initial_p, lambda, args.lo_p, args.hi_p, sigma, nullptr, seed, false), N(N_)
    {
        opt.init(N_);
    }
};

typedef Handle<LMCMA> HL;
typedef Handle<SepCMA> HS;
typedef Handle<CMAChol> HC;

template <typename H>
void tell_all(H* h, const double* fitness)
{
    std::vector<double> x(h->N);
    for (int i = 0; i < h->opt.lambda; ++i) {
        double f = fitness[i];
        h->opt.getNextParameterVector(x.data(), h->N);
        h->opt.setEvaluationFeedback(&f, 1);
    }
}

}  // namespace

extern "C" {

// ---- primitives (lmcma.cpp:14-104, 812-834) -------------------------------------------------
void ref_uniform_stream(long seed, int n, double* out)
{
    random_t r;
    random_init(&r, (unsigned long)seed);
    for (int i = 0; i < n; ++i) out[i] = random_Uniform(&r);
    random_exit(&r);
}

void ref_gauss_stream(long seed, int n, double* out)
{
    random_t r;
    random_init(&r, (unsigned long)seed);
    for (int i = 0; i < n; ++i) out[i] = random_Gauss(&r);
    random_exit(&r);
}

void ref_sort(int n, double* values, int* index)
{
    std::vector<sortedvals> tmp(n);
    myqsort(n, values, index, tmp.data());
}

void ref_differentiation_matrix(int n, int order, double dt, double* out, int row_len)
{
    differentiationMatrix(n, order, dt, out, row_len);
}

// ---- LMCMA ------------------------------------------------------------------------------
void* ref_lmcma_create(int N, int lambda, const double* initial, const double* lo, const double* hi, double sigma,
                       int seed)
{
    return new HL(N, lambda, initial, lo, hi, sigma, seed);
}
void ref_lmcma_destroy(void* h) { delete (HL*)h; }
void ref_lmcma_ask(void* h, double* x) { ((HL*)h)->opt.getNextParameterVector(x, ((HL*)h)->N); }
void ref_lmcma_tell(void* h, double* f, int nf) { ((HL*)h)->opt.setEvaluationFeedback(f, nf); }
void ref_lmcma_tell_all(void* h, const double* fitness) { tell_all((HL*)h, fitness); }
int ref_lmcma_done(void* h) { return ((HL*)h)->opt.isBehaviorLearningDone() ? 1 : 0; }

#define REF_INT(name, expr) int ref_lmcma_##name(void* h) { LMCMA& o = ((HL*)h)->opt; return expr; }
#define REF_DBL(name, expr) double ref_lmcma_##name(void* h) { LMCMA& o = ((HL*)h)->opt; return expr; }
#define REF_PTR(type, name, expr) const type* ref_lmcma_##name(void* h) { LMCMA& o = ((HL*)h)->opt; return expr; }
REF_INT(lambda, o.lambda)
REF_INT(mu, o.mu)
REF_INT(nvectors, o.nvectors)
REF_INT(itr, o.itr)
REF_INT(iterator_sz, o.iterator_sz)
REF_INT(counteval, o.counteval)
REF_DBL(bestf, o.BestF)
REF_DBL(sigma, o.sigma)
REF_DBL(s, o.s)
REF_DBL(mueff, o.mueff)
REF_DBL(c1, o.c1)
REF_DBL(cc, o.cc)
REF_DBL(K, o.K)
REF_DBL(M, o.M)
REF_PTR(double, arx, o.arx)
REF_PTR(double, arfitness, o.arfitness)
REF_PTR(double, prev_arfitness, o.prev_arfitness)
REF_PTR(double, xmean, o.xmean)
REF_PTR(double, pc, o.pc)
REF_PTR(double, weights, o.weights)
REF_PTR(double, v_arr, o.v_arr)
REF_PTR(double, pc_arr, o.pc_arr)
REF_PTR(double, nj, o.Nj_arr)
REF_PTR(double, lj, o.Lj_arr)
REF_PTR(int, iterator, o.iterator)
REF_PTR(int, arindex, o.arindex)
REF_PTR(int, t, o.t)
REF_PTR(int, vec, o.vec)

// ---- SepCMA (lmcma.cpp:466-577) ------------------------------------------------------------
void* ref_sep_create(int N, int lambda, const double* initial, const double* lo, const double* hi, double sigma, int seed)
{
    return new HS(N, lambda, initial, lo, hi, sigma, seed);
}
void ref_sep_destroy(void* h) { delete (HS*)h; }
void ref_sep_tell_all(void* h, const double* fitness) { tell_all((HS*)h, fitness); }
int ref_sep_lambda(void* h) { return ((HS*)h)->opt.lambda; }
double ref_sep_sigma(void* h) { return ((HS*)h)->opt.sigma; }
const double* ref_sep_arx(void* h) { return ((HS*)h)->opt.arx; }
const double* ref_sep_xmean(void* h) { return ((HS*)h)->opt.xmean; }
const double* ref_sep_diag(void* h) { return ((HS*)h)->opt.diagD; }

// ---- CMAChol (lmcma.cpp:579-754) -----------------------------------------------------------
void* ref_chol_create(int N, int lambda, const double* initial, const double* lo, const double* hi, double sigma, int seed)
{
    return new HC(N, lambda, initial, lo, hi, sigma, seed);
}
void ref_chol_destroy(void* h) { delete (HC*)h; }
void ref_chol_tell_all(void* h, const double* fitness) { tell_all((HC*)h, fitness); }
int ref_chol_lambda(void* h) { return ((HC*)h)->opt.lambda; }
double ref_chol_sigma(void* h) { return ((HC*)h)->opt.sigma; }
const double* ref_chol_arx(void* h) { return ((HC*)h)->opt.arx; }
const double* ref_chol_xmean(void* h) { return ((HC*)h)->opt.xmean; }
const double* ref_chol_factor(void* h) { return ((HC*)h)->opt.A; }

// ---- the reference's own loop, timed (the caller's loop of example_lmcma.cpp:49-60) ----------
// cost(user, row, N) scores one candidate; returns wall seconds for `generations` generations of
// ask / score / tell through the reference's per-candidate interface.  Single thread, like the
// reference.
typedef double (*ref_cost_fn)(void* user, const double* row, int N);

double ref_lmcma_run_timed(void* h, ref_cost_fn cost, void* user, int generations)
{
    HL* H = (HL*)h;
    std::vector<double> x(H->N);
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const long evals = (long)generations * H->opt.lambda;
    for (long e = 0; e < evals; ++e) {
        H->opt.getNextParameterVector(x.data(), H->N);
        double f = cost(user, x.data(), H->N);
        H->opt.setEvaluationFeedback(&f, 1);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

}  // extern "C"
