// SepCMA and CMAChol through include/lmcma.hpp, the way a caller of the reference's header uses
// them (reference lmcma.hpp:152-239): ask/tell on a rotated quadratic.  Usage: sibling_optimisers [seed]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "lmcma.hpp"

static const int kN = 12;

static double rotated_quadratic(const double *x)
{
    // sum_i 10^(2 i / (N - 1)) y_i^2 with y = pairwise rotations of x - 1
    double f = 0;
    for (int i = 0; i < kN; i += 2) {
        const double a = x[i] - 1.0, b = x[i + 1] - 1.0;
        const double y0 = 0.8 * a + 0.6 * b, y1 = -0.6 * a + 0.8 * b;
        f += std::pow(10.0, 2.0 * i / (kN - 1)) * y0 * y0 + std::pow(10.0, 2.0 * (i + 1) / (kN - 1)) * y1 * y1;
    }
    return f;
}

template <class Optimiser>
static double run(int seed, int evaluations)
{
    double start[kN], x[kN];
    for (int i = 0; i < kN; ++i) start[i] = 3.0;
    Optimiser opt(start, 16, 0, 0, 1.0, 0, seed);
    opt.init(kN);
    for (int i = 0; i < evaluations; ++i) {
        opt.getNextParameterVector(x, kN);
        double f = rotated_quadratic(x);
        opt.setEvaluationFeedback(&f, 1);
    }
    return opt.BestF;
}

int main(int argc, char **argv)
{
    const int seed = argc > 1 ? std::atoi(argv[1]) : 1;
    const double sep = run<SepCMA>(seed, 16 * 400), chol = run<CMAChol>(seed, 16 * 400);
    std::printf("SepCMA best %.3e CMAChol best %.3e\n", sep, chol);
    return (sep < 1e-3 && chol < 1e-6) ? 0 : 1;
}
