"""Generation rate of the drop-in class in its host-driven modes at the headline shape
(lambda = 4096, N = 128, m = 16), next to the device-resident loop of bench.py.

    python tools/host_mode_rate.py

Mode 1: the reference's sequential normals drawn on the host and uploaded each generation (2 MB),
        device sample + score + rank + recombine, host fp64 update.
Mode 2: Philox normals on the device, host fp64 update (no upload of normals).
Mode 3: everything on the device (the bench.py path), through the class."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lmcma_path_planner_amd import LMCMA, grids

prob = grids.benchmark_problem(2)
for name, philox, dev_update, gens in (("host normals + host update", False, False, 30),
                                       ("device normals + host update", True, False, 100),
                                       ("device-resident", True, True, 2000)):
    opt = LMCMA(prob.x0, prob.lam, prob.lo, prob.hi, prob.sigma0, None, 42)
    opt.setOptions(m=prob.m, philox=philox, device_update=dev_update)
    opt.setTrajectoryProblem(2, prob.W, prob.occ, prob.start, prob.goal)
    opt.init(prob.N)
    opt.step(5)
    t0 = time.perf_counter()
    opt.step(gens)
    dt = (time.perf_counter() - t0) / gens
    print("%-30s %9.1f us per generation  %.3g candidate-trajectories/s" % (name, dt * 1e6, prob.lam / dt), flush=True)
    opt.close()
