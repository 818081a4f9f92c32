#!/usr/bin/env python3
"""The byte-moving kernel alone over population sizes:  python tools/generate_sweep.py [cfg]
k_generate (sample + score in one launch: writes lambda rows and costs, reads the direction store and
the grid) timed back to back on one stream (events around `reps` launches), cfg2's geometry by
default.  Algorithmic bytes per launch = lambda (4 N + 4 + 12) + 8 m N + grid bytes; the fraction is
against the nominal 8 TB/s.  At the benchmark's 4096 candidates a launch is start-up and latency; the
sweep shows what the same kernel moves when there is enough of it."""
import os
import sys

sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob = grids.benchmark_problem(cfg)
print("cfg%d geometry: N = %d, m = %d, grid %s" % (cfg, prob.N, prob.m, "x".join(str(s) for s in prob.cells.shape)))
for lam in ((4096, 16384, 65536, 262144, 1048576) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2:]]):
    c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start, goal=prob.goal)
    c.set_grid(prob.cells)
    c.set_bounds(prob.lo, prob.hi)
    c.opt_init(prob.x0, prob.sigma0)
    reps = max(20, min(2000, (1 << 24) // lam))

    def timed():
        for _ in range(5):
            c.phase_local()
        c.synchronize()
        c.stream_mark(0)
        for _ in range(reps):
            c.phase_local()
        c.stream_mark(1)
        c.synchronize()
        return c.stream_elapsed() * 1e3 / reps

    us_empty = timed()                  # an empty direction store: normals, bounds and scoring only
    c.run(prob.m + 4)                   # a full direction store: the kernel's steady-state work
    us = timed()
    bytes_ = lam * (4 * prob.N + 4 + 12) + 8 * prob.m * prob.N + prob.grid_bytes()
    print("lambda = %8d: %9.2f us per launch, %7.1f MB, %7.1f GB/s algorithmic = %.3f of 8 TB/s, %.3e candidates/s"
          % (lam, us, bytes_ / 1e6, bytes_ / us / 1e3, bytes_ / us / 1e3 / 8000.0, lam / us * 1e6), flush=True)
    print("                   with an empty direction store %9.2f us: the %d direction vectors cost %.2f us" % (us_empty, prob.m, us - us_empty))
    c.close()
