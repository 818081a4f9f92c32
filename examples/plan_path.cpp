// Batched planning on the two-bar map: the population is sampled, scored, ranked and
// recombined on the GPU; step() replaces lambda ask/tell round trips per generation.
// Usage: plan_path [size] [waypoints] [lambda] [m] [generations]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lmcma.hpp"
#include "lmcma_opt.h"

int main(int argc, char **argv)
{
    const int size = argc > 1 ? std::atoi(argv[1]) : 100;
    const int W = argc > 2 ? std::atoi(argv[2]) : 32;
    const int lambda = argc > 3 ? std::atoi(argv[3]) : 64;
    const int m = argc > 4 ? std::atoi(argv[4]) : 8;
    const int generations = argc > 5 ? std::atoi(argv[5]) : 500;
    const int N = 2 * W;

    std::vector<uint8_t> occ((size_t)size * size);
    lmcma_grid_two_bar(size, 1, occ.data());

    LMCMATrajectoryProblem problem;
    problem.dims = 2;
    problem.waypoints = W;
    problem.nx = size;
    problem.ny = size;
    problem.cells = occ.data();
    problem.start[0] = size - 1.0; problem.start[1] = 0.0;   // the reference's (99,0) -> (0,99)
    problem.goal[0] = 0.0;         problem.goal[1] = size - 1.0;

    std::vector<double> x0(N), lo(N, 0.0), hi(N, size - 1.0);
    for (int k = 0; k < W; ++k) {
        const double t = (k + 1.0) / (W + 1.0);
        x0[k] = problem.start[0] + t * (problem.goal[0] - problem.start[0]);
        x0[W + k] = problem.start[1] + t * (problem.goal[1] - problem.start[1]);
    }

    LMCMA opt(x0.data(), lambda, lo.data(), hi.data(), size / 20.0, 0, 42);
    opt.setDirectionVectors(m);
    opt.usePhiloxSampling(true);
    opt.useDeviceUpdate(true);
    opt.setTrajectoryProblem(problem);
    opt.init(N);
    opt.step(generations);

    std::vector<double> path(N);
    opt.mean(path.data());
    std::printf("generations %d evaluations %d best cost %.3f sigma %.3g\n", opt.generation(), opt.counteval,
                opt.BestF, opt.stepSize());
    for (int k = 0; k < W; k += W / 8 > 0 ? W / 8 : 1) std::printf("  wp %2d: (%.2f, %.2f)\n", k, path[k], path[W + k]);
    return 0;
}
