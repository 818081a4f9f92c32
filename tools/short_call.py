"""What a short timed region costs around its one library call:  python tools/short_call.py [n]
The region of bench.py --steps n (generations 45 .. 45 + n of an episode, as the driver's
--steps 20 --warmup 5 times them) under different brackets: with and without the stream marks,
ended by the library's stream synchronisation, by torch.cuda.synchronize(), or by both."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from lmcma_path_planner_amd import grids, hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prob = grids.benchmark_problem(2)
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells)
ctx.set_bounds(prob.lo, prob.hi)
ctx.opt_init(prob.x0, prob.sigma0)
ctx.run(200)
ctx.synchronize()


def trial(marks, lib_sync, torch_sync):
    ctx.opt_reset()
    ctx.run(45)
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if marks:
        ctx.stream_mark(0)
    ctx.run(n)
    t1 = time.perf_counter()
    if marks:
        ctx.stream_mark(1)
    if lib_sync:
        ctx.synchronize()
    t2 = time.perf_counter()
    if torch_sync:
        torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6, (ctx.stream_elapsed() * 1e3 if marks else 0.0)


def med(v):
    return sorted(v)[len(v) // 2]


for name, cfg in (("marks, library sync, torch sync (bench.py)", (True, True, True)),
                  ("no marks, library sync, torch sync", (False, True, True)),
                  ("no marks, torch sync only", (False, False, True)),
                  ("no marks, library sync only", (False, True, False))):
    first = [trial(*cfg)[2] for _ in range(12)]
    r = [trial(*cfg) for _ in range(40)]
    print("   first twelve trials, all done: %s us" % ", ".join("%.1f" % v for v in first))
    q, a, b, d = (med([x[i] for x in r]) for i in range(4))
    print("%-44s call returns %.1f us, library sync done %.1f, all done %.1f us = %.2f per generation; between marks %.1f"
          % (name, q, a, b, b / n, d), flush=True)
