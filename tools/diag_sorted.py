"""Phase stamps of the one-workgroup ranking of a batch (diagnostic build), query 0, ns.
    make -C lmcma_path_planner_amd/csrc diag && python tools/diag_sorted.py [Q]"""
import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip, _lib

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = grids.benchmark_problem(2)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42,
                     queries=Q)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
names = ["loads", "samples", "buckets", "prefix+place", "in-bucket", "to LDS in order", "out + success rule"]
for g in range(40):
    ctx.run(1)
    sr = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 1, sr)
    if g in (0, 1, 5, 20, 39):
        print(g, {names[i]: (int(sr[9 + i]) - int(sr[8 + i])) * 10 for i in range(7)}, "total", (int(sr[15]) - int(sr[8])) * 10, flush=True)
