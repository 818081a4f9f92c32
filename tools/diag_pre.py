"""Phase times of the update's first part when it rides in the sampling launch (diagnostic build:
make -C lmcma_path_planner_amd/csrc diag).  python tools/diag_pre.py [cfg]"""
import os, sys, ctypes
os.environ["LMCMA_HIP_LIB"] = os.path.join(os.getcwd(), "lmcma_path_planner_amd/lib/liblmcma_hip_diag.so")
os.environ["LMCMA_HIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip, _lib

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prob = grids.benchmark_problem(cfg)
lam = prob.lam if cfg <= 3 else prob.lam // 8
ctx = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
lib = _lib.load()
lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
for g in range(70):
    ctx.run(1)
    su = (ctypes.c_uint64 * 16)()
    lib.lmcma_hip_diag_stamps(ctx._h, 2, su)
    if g >= 33 and g % 4 == 0:
        imin, isz = su[12] >> 8, su[12] & 255
        print("gen %d isz %d imin %d: loads %d ns, publish untouched %d, steps %d (%.0f ns per step), whole %d" % (
            g, isz, imin, (su[9] - su[8]) * 10, (su[10] - su[9]) * 10, (su[11] - su[10]) * 10,
            (su[11] - su[10]) * 10 / max(isz - 2, 1), (su[11] - su[8]) * 10), flush=True)
