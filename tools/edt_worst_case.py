import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lmcma_path_planner_amd import hip
for shape in ((1024, 1024), (256, 256, 256)):
    dims = len(shape)
    occ = np.zeros(shape, dtype=np.uint8); occ.flat[0] = 1
    ctx = hip.HipContext(dims * 8, 64, 4, dims=dims, waypoints=8, start=[0.0] * dims, goal=[1.0] * dims)
    ctx.set_grid_from_occupancy(occ)
    t0 = time.perf_counter(); d = ctx.set_grid_from_occupancy(occ, want_distances=True); dt = time.perf_counter() - t0
    idx = np.indices(shape).astype(np.float64)
    want = np.sqrt((idx ** 2).sum(axis=0)).astype(np.float32)
    print(shape, "one occupied corner cell: %.1f ms, exact %s" % (dt * 1e3, bool((d == want).all())), flush=True)
    ctx.close()
