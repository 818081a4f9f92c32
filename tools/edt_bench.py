"""Time the device distance transform on the benchmark grids.

    python tools/edt_bench.py            (under rocprofv3 --kernel-trace --stats for per-kernel times)

Prints wall time of lmcma_hip_set_grid_from_occupancy (upload, three passes, finish, two
4-byte reads) for the 1024^2 and 256^3 maps, and the host transform beside it."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lmcma_path_planner_amd import grids, hip

for cfg in (2, 4, 5):
    prob = grids.benchmark_problem(cfg)
    occ = prob.occ
    ctx = hip.HipContext(prob.N, 256, 4, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal)
    ctx.set_grid_from_occupancy(occ)       # first call: module load, allocations
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.set_grid_from_occupancy(occ)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    grids.edt(occ)
    host = time.perf_counter() - t0
    print("%s: %s cells, %.1f%% occupied: device %.2f ms per call (with upload), host transform %.1f ms"
          % (prob.name, "x".join(map(str, occ.shape)), 100.0 * occ.mean(), dt * 1e3, host * 1e3), flush=True)
    ctx.close()
