import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from lmcma_path_planner_amd import grids, hip
import test_gpu_sharded_multi as T
prob = grids.benchmark_problem(2)
for lam in (4096, 32768):
    a, b, c = T.make_whole(prob, lam, 3)
    for g in range(6):
        a.run(1)
        b.phase_local(); b.phase_rank(); b.phase_update(0)
        c.phase_local_range(0, lam // 2); c.phase_local_range(lam // 2, lam // 2); c.phase_rank(); c.phase_update(0)
        sa, sb, sc = a.opt_get(), b.opt_get(), c.opt_get()
        print(lam, g, "sigma", sa["sigma"], sb["sigma"], sc["sigma"], "s", sa["s"], sb["s"], "best", sa["best_f"], sb["best_f"],
              "xmean equal", np.array_equal(sa["xmean"], sb["xmean"]), np.array_equal(sa["xmean"], sc["xmean"]),
              "costs equal", np.array_equal(a.costs(), b.costs()), np.array_equal(a.costs(), c.costs()))
    for x in (a, b, c): x.close()
