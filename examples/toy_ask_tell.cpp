// Ask/tell on the two-bump toy function, written against include/lmcma.hpp the way a
// caller of the reference's lmcma.hpp would write it (the reference's only usage example
// is lmcma_path_planner/src/example_lmcma.cpp:28-76; this file is an independent driver
// for the same API, not a copy of it).  Usage: toy_ask_tell [seed] [evaluations]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "lmcma.hpp"

static double two_bumps(double x, double y)
{
    // -(4 exp(-|p-(4,4)|^2) + 2 exp(-|p-(2,2)|^2)): global minimum -4.0007 near (4,4)
    const double a = (x - 4) * (x - 4) + (y - 4) * (y - 4);
    const double b = (x - 2) * (x - 2) + (y - 2) * (y - 2);
    return -(4.0 * std::exp(-a) + 2.0 * std::exp(-b));
}

int main(int argc, char **argv)
{
    const int seed = argc > 1 ? std::atoi(argv[1]) : 1;
    const int evaluations = argc > 2 ? std::atoi(argv[2]) : 1000;
    double start[2] = {0.0, 0.0};
    double lo[2] = {-2.0, -2.0}, hi[2] = {15.0, 15.0};
    double x[2];

    LMCMA opt(start, -1, lo, hi, 1.0, 0, seed);  // lambda < 1: 4 + int(3 ln N)
    opt.init(2);
    CMABase *base = &opt;   // every second evaluation goes through the base class, as a caller holding a CMABase* would
    for (int i = 0; i < evaluations && !opt.isBehaviorLearningDone(); ++i) {
        if (i & 1) base->getNextParameterVector(x, 2);
        else opt.getNextParameterVector(x, 2);
        double f = two_bumps(x[0], x[1]);
        opt.setEvaluationFeedback(&f, 1);
    }
    std::printf("evaluations %d best %.6f last (%.4f, %.4f) sigma %.3g\n", opt.counteval, opt.BestF, x[0], x[1],
                opt.stepSize());
    return (opt.BestF < -1.9) ? 0 : 1;
}
