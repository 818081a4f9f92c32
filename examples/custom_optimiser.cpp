// An optimiser derived from CMABase the way the reference's own variants are (lmcma.hpp:89, 152,
// 199): it uses the protected state (arx, arfitness, xmean, weights, mu, lambda, N), the base
// class's sampleStandardNormal / applyBoundaries and its ask/tell calls.  Runs on the host; no
// GPU is touched.  Usage: custom_optimiser [seed]
// Prints the first candidate (17 digits), then evaluations and best value on a shifted sphere.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lmcma.hpp"

class WeightedES : public CMABase {
    double sigma;
    std::vector<int> order;
    std::vector<sortedvals> scratch;
    std::vector<double> z;

public:
    WeightedES(double *initial, int lambda, double *lo, double *hi, double sigma0, int seed)
        : CMABase(initial, lambda, lo, hi, 0, seed), sigma(sigma0) {}

    void init(int n)
    {
        CMABase::init(n);
        order.resize(lambda);
        scratch.resize(lambda);
        z.resize(N);
        sample();
    }

    void sample()
    {
        for (int i = 0; i < lambda; ++i) {
            sampleStandardNormal(z.data());
            for (int n = 0; n < N; ++n) arx[i * N + n] = xmean[n] + sigma * z[n];
        }
        applyBoundaries();
    }

    void update()
    {
        myqsort(lambda, arfitness, order.data(), scratch.data());
        for (int n = 0; n < N; ++n) {
            xold[n] = xmean[n];
            double acc = 0.0;
            for (int i = 0; i < mu; ++i) acc += weights[i] * arx[order[i] * N + n];
            xmean[n] = acc;
        }
        sigma *= 0.97;
        ++itr;
    }

    bool isBehaviorLearningDone() { return sigma < 1e-8; }
    int populationSize() const { return lambda; }
};

int main(int argc, char **argv)
{
    const int seed = argc > 1 ? std::atoi(argv[1]) : 7;
    const int N = 6;
    double x0[N] = {3.0, -2.0, 1.0, 0.5, -4.0, 2.5};
    double lo[N], hi[N];
    for (int n = 0; n < N; ++n) { lo[n] = -5.0; hi[n] = 3.2; }
    WeightedES es(x0, 0, lo, hi, 0.8, seed);
    CMABase *opt = &es;            // callers of the reference hold the base class
    opt->init(N);
    std::vector<double> x(N);
    opt->getNextParameterVector(x.data(), N);
    std::printf("lambda %d first", es.populationSize());
    for (int n = 0; n < N; ++n) std::printf(" %.17g", x[n]);
    std::printf("\n");
    for (int e = 0; e < 6000 && !opt->isBehaviorLearningDone(); ++e) {
        opt->getNextParameterVector(x.data(), N);
        double f = 0.0;
        for (int n = 0; n < N; ++n) f += (x[n] - 1.0) * (x[n] - 1.0);
        double parts[2] = {0.25 * f, 0.75 * f};    // several feedbacks are summed (lmcma.cpp:186-188)
        opt->setEvaluationFeedback(parts, 2);
    }
    std::printf("evaluations %d best %.6g\n", opt->counteval, opt->BestF);
    return 0;
}
