"""Host time of the first lmcma_hip_run call on a context (one eager generation, then the
graphs are recorded and instantiated) and of the calls after it:  python tools/first_call_time.py [cfg]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob = grids.benchmark_problem(cfg)
lam = prob.lam if cfg <= 3 else prob.lam // 8
for trial in range(2):
    ctx = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
    ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
    ctx.synchronize()
    for n in (1, 5, 20, 100):
        t0 = time.perf_counter(); ctx.run(n); ctx.synchronize(); dt = time.perf_counter() - t0
        print("context %d: run(%d) %.3f ms (%.1f us per generation)" % (trial, n, dt * 1e3, dt * 1e6 / n), flush=True)
