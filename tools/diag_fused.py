"""Phase stamps of the sampling launch that carries the previous generation's update (diagnostic build):
ns from the start of the launch's last workgroup.
    make -C lmcma_path_planner_amd/csrc diag && python tools/diag_fused.py"""
import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip, _lib

prob = grids.benchmark_problem(2)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
for g in (10, 20, 30, 40):
    ctx.run(g)   # the last sampling launch of the call carried an update
    sg = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 2, sg)
    t0 = int(sg[0])
    rel = lambda i: (int(sg[i]) - t0) * 10
    print("update in the launch: start %d, loads issued %d, rows summed %d, mean+path %d, newest vector %d, stores %d | "
          "newest vector's group: from %d to %d | sampling: normals %d, barrier %d, directions %d, rows out %d, scored %d, end %d"
          % (rel(8), rel(9), rel(10), rel(11), rel(12), rel(13), rel(14), rel(15), rel(1), rel(2), rel(3), rel(4), rel(5), rel(6)), flush=True)
