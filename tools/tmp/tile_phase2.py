import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from lmcma_path_planner_amd import grids, hip
import test_gpu_sharded_multi as T
prob = grids.benchmark_problem(2)
for lam in (16384, 32768, 65536):
    a, b, c, d = T.make_whole(prob, lam, 4)
    a.run(20)
    for g in range(20):
        b.phase_local(); b.phase_rank(); b.phase_update(0)
        c.run(1)
    d.run(1); d.run(10); d.run(9)
    sa, sb, sc, sd = a.opt_get(), b.opt_get(), c.opt_get(), d.opt_get()
    print(lam, "run(20)", sa["sigma"], "phases", sb["sigma"], "run(1)x20", sc["sigma"], "run 1+10+9", sd["sigma"], flush=True)
    for x in (a, b, c, d): x.close()
