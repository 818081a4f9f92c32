import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from lmcma_path_planner_amd import grids, hip
import test_gpu_sharded_multi as T
prob = grids.benchmark_problem(2)
lam, G = 32768, 4
for trial in range(2):
    one = T.make_whole(prob, lam, 1)[0]
    ctxs = T.make_whole(prob, lam, G)
    for g in range(4):
        one.run(1)
        hip.HipContext.run_sharded_group(ctxs, 1)
        for c in ctxs: c.synchronize()
        so = one.opt_get()
        R, C = one.rows(), one.costs()
        for r, c in enumerate(ctxs):
            s = c.opt_get()
            rr, cc = c.rows(), c.costs()
            bad_rows = np.nonzero((rr != R).any(axis=1))[0]
            bad_cost = np.nonzero(cc != C)[0]
            print("trial", trial, "gen", g, "ctx", r, "sigma", s["sigma"] == so["sigma"], "xmean", np.array_equal(s["xmean"], so["xmean"]),
                  "rows differing", len(bad_rows), bad_rows[:4], bad_rows[-4:] if len(bad_rows) else "", "costs differing", len(bad_cost), bad_cost[:4], flush=True)
    one.close()
    for c in ctxs: c.close()
