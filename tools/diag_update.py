"""Phase timing inside the ranking kernel and its update tail (diagnostic build only).

Build the diagnostic library first:
    make -C lmcma_path_planner_amd/csrc diag
then run on a GPU:  python tools/diag_update.py
"""
import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip, _lib

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob = grids.benchmark_problem(cfg)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
rank_names = ["stage", "count", "psr+scatter", "recombine"]
upd_names = ["loads+row sums", "tables+publish", "mean+path", "vectors", "stores"]
for g in range(60):
    ctx.run(1)
    sr = (ctypes.c_uint64 * 16)()
    su = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 1, sr)
    lib.lmcma_hip_diag_stamps(ctx._h, 0, su)
    sg = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 2, sg)
    if g in (0, 1, 4, 8, 15, 20, 30, 40, 59):
        rank = {rank_names[i]: (sr[9 + i] - sr[8 + i]) * 10 for i in range(4)}
        upd = {upd_names[i]: (su[1 + i] - su[i]) * 10 for i in range(5)}
        print(g, "rank (last workgroup), ns:", rank, "total", (sr[12] - sr[8]) * 10,
              "| update:", upd, "total", (su[5] - su[0]) * 10, flush=True)
        gen_names = ["loads+philox", "barrier", "directions", "row out+barrier", "score", "cost out"]
        print("   generate (last workgroup), ns:", {gen_names[i]: (sg[1 + i] - sg[i]) * 10 for i in range(6)},
              "total", (sg[6] - sg[0]) * 10, flush=True)
