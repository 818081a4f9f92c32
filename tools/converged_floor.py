"""How far the device loop and the fp64 oracle get on the obstacle map whose optimum lies in free space
(tests/test_gpu_end_to_end.py), generation by generation:  python tools/converged_floor.py
Prints J - J(straight line) and the step size of both after 500 .. 8000 generations for 12 seeds."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from lmcma_path_planner_amd import grids, hip
from oracle import pyoracle as oracle
import test_gpu_end_to_end as E
prob = grids.benchmark_problem(1)
occ = E._free_diagonal_map(100, 44)
t = np.arange(1, prob.W + 1) / (prob.W + 1)
x0 = prob.x0 - np.concatenate([14 * np.sin(np.pi * t), 14 * np.sin(np.pi * t)])
co = oracle.CostOracle(2, prob.W, occ, prob.start, prob.goal)
jstar, _ = co.cost(prob.x0)
print("straight line cost", jstar)
lam, m = 32, 8
oracle.set_threads(8)
for seed in range(5, 17):
    o = oracle.OracleLMCMA(prob.N, lam, m, initial=x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=seed, z_source=oracle.Z_PHILOX)
    ctx = hip.HipContext(prob.N, lam, m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=seed)
    ctx.set_grid(occ); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(x0, prob.sigma0)
    for gens in (500, 500, 1000, 2000, 4000):
        o.run_trajectory(co, gens)
        ctx.run(gens)
        st = ctx.opt_get()
        jc, jg = co.cost(o.xmean)[0], co.cost(st["xmean"])[0]
        print("seed %d itr %5d  oracle J-J* %.3e sigma %.2e | device J-J* %.3e sigma %.2e best_f-J* %.3e"
              % (seed, st["itr"], jc - jstar, o.sigma, jg - jstar, st["sigma"], st["best_f"] - jstar), flush=True)
    ctx.close()
