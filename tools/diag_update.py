import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
sys.path.insert(0, os.getcwd())
import numpy as np
from lmcma_path_planner_amd import grids, hip, _lib
prob = grids.benchmark_problem(2)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.occ); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_update_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
acc = {}
for g in range(60):
    ctx.run(1)
    st = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_update_stamps(ctx._h, st)
    d = [(st[i + 1] - st[i]) * 10 for i in range(7)]  # ns
    d.append(("S1 loads", (st[8] - st[2]) * 10, "shuffles", (st[9] - st[8]) * 10, "rest", (st[3] - st[9]) * 10))
    d.append("clock MHz %.0f" % ((st[11] - st[10]) / ((st[7] - st[0]) * 10e-3)))
    if g in (0, 1, 4, 8, 15, 20, 30, 40, 59):
        print(g, "ns per phase:", d, "total", (st[7] - st[0]) * 10)
