// CMABase of include/lmcma.hpp: the reference's base class (lmcma_path_planner/src/lmcma.hpp:41-81,
// lmcma.cpp:107-230, cited as L/) for optimisers a caller derives from it.  Host code only; the
// library's own LMCMA / SepCMA / CMAChol override the two hooks and keep their populations on
// the GPU.
#include <math.h>

#include <algorithm>
#include <iostream>
#include <limits>

#include "lmcma.hpp"

CMABase::CMABase(double *initialParams_, int lambda_, double *loBounds_, double *hiBounds_, double *covariance_,
                 int inseed_, bool verbose_)
    : N(0), lambda(lambda_), mu(0), mueff(0.0), initialParams(initialParams_), loBounds(loBounds_),
      hiBounds(hiBounds_), arx(0), arfitness(0), xmean(0), xold(0), weights(0), covariance(covariance_),
      inseed(inseed_), verbose(verbose_), L(0), itr(0), sampleIdx(0), counteval(0),
      BestF(std::numeric_limits<double>::max())
{
    random_init(&ttime, (long unsigned)(inseed_ < 1 ? 0 : inseed_));   // L/:114; a seed < 1 means the clock (L/:40-45)
}

CMABase::~CMABase()
{
    random_exit(&ttime);
    delete[] arx;
    delete[] arfitness;
    delete[] xmean;
    delete[] xold;
    delete[] weights;
    delete[] L;
}

// L/:130-170: population size, recombination weights, initial mean, factor of a given covariance
void CMABase::init(int n)
{
    N = n;
    if (lambda < 1) lambda = 4 + int(3 * log((double)N));
    mu = lambda / 2;
    delete[] arx; delete[] arfitness; delete[] xmean; delete[] xold; delete[] weights; delete[] L;
    L = 0;
    arx = new double[(size_t)N * lambda];
    arfitness = new double[lambda];
    xmean = new double[N];
    xold = new double[N];
    weights = new double[mu > 0 ? mu : 1];
    double total = 0.0;
    for (int i = 0; i < mu; ++i) {
        weights[i] = log(double(mu) + 0.5) - log(double(1 + i));
        total += weights[i];
    }
    double sq = 0.0;
    for (int i = 0; i < mu; ++i) {
        weights[i] /= total;
        sq += weights[i] * weights[i];
    }
    mueff = 1.0 / sq;
    for (int i = 0; i < N; ++i) xmean[i] = initialParams ? initialParams[i] : random_Uniform(&ttime);
    if (covariance) {
        L = new double[(size_t)N * N];
        cholesky(covariance, L, N);
    }
}

void CMABase::getNextParameterVector(double *params, int n) { askHook(params, n); }
void CMABase::setEvaluationFeedback(double *feedbacks, int numFeedbacks) { tellHook(feedbacks, numFeedbacks); }
bool CMABase::isBehaviorLearningDone() { return false; }

// L/:172-182
void CMABase::askHook(double *params, int n)
{
    const double *row = arx + (size_t)sampleIdx * n;
    std::copy(row, row + n, params);
}

// L/:184-205
void CMABase::tellHook(double *feedbacks, int numFeedbacks)
{
    double feedback = 0.0;
    for (int i = 0; i < numFeedbacks; ++i) feedback += feedbacks[i];
    arfitness[sampleIdx] = feedback;
    ++counteval;
    if (feedback < BestF || counteval == 1) {
        BestF = feedback;
        if (verbose) std::cout << "Functions evaluation #" << counteval << ", value: " << feedback << std::endl;
    }
    if (++sampleIdx % lambda == 0) {
        update();
        sample();
        sampleIdx = 0;
    }
}

// L/:212-218
void CMABase::sampleStandardNormal(double *z)
{
    for (int n = 0; n < N; ++n) z[n] = random_Gauss(&ttime);
    if (L) applyCovL(L, z, N);
}

// L/:220-230: the lower bounds first, then the upper ones
void CMABase::applyBoundaries()
{
    const size_t total = (size_t)lambda * N;
    if (loBounds)
        for (size_t e = 0; e < total; ++e) arx[e] = std::max(arx[e], loBounds[e % N]);
    if (hiBounds)
        for (size_t e = 0; e < total; ++e) arx[e] = std::min(arx[e], hiBounds[e % N]);
}
