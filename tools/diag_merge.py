"""Phase timing inside the merge of sorted tiles / shards, k_rank_merge (diagnostic build only).

    make -C lmcma_path_planner_amd/csrc diag
    python tools/diag_merge.py [cfg]      (5: 16384 candidates in one context = four tiles)
"""
import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip, _lib

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
prob = grids.benchmark_problem(cfg)
lam = prob.lam if cfg <= 3 else prob.lam // 8
ctx = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
names = ["own key + pivots", "searches + windows", "publish", "elite rows + stores"]
for g in range(40):
    ctx.run(1)
    sr = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 1, sr)
    if g in (0, 1, 4, 8, 15, 20, 30, 39):
        print(g, "merge (last workgroup), ns:", {names[i]: (sr[9 + i] - sr[8 + i]) * 10 for i in range(4)},
              "total", (sr[12] - sr[8]) * 10, "| pivot search (last round)", (sr[13] - sr[9]) * 10, flush=True)
