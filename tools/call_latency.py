"""Where the fixed cost of a short lmcma_hip_run call goes:  python tools/call_latency.py [n] [LMCMA_HIP_GRAPH_GENS]
Host time of the call itself (queueing), wall time to completion, and the device time between
stream marks, for calls of n generations on an idle stream."""
import sys, time, os
if len(sys.argv) > 2:
    os.environ["LMCMA_HIP_GRAPH_GENS"] = sys.argv[2]
sys.path.insert(0, os.getcwd())
from lmcma_path_planner_amd import grids, hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prob = grids.benchmark_problem(2)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
ctx.run(50); ctx.synchronize()
q, w, d = [], [], []
marks = os.environ.get("CALL_MARKS", "1") != "0"   # CALL_MARKS=0: no stream marks around the call
for trial in range(30):
    ctx.opt_reset()     # every trial the same n generations of a fresh query
    ctx.synchronize()
    t0 = time.perf_counter()
    if marks: ctx.stream_mark(0)
    ctx.run(n)
    t1 = time.perf_counter()
    if marks: ctx.stream_mark(1)
    ctx.synchronize()
    t2 = time.perf_counter()
    q.append((t1 - t0) * 1e6); w.append((t2 - t0) * 1e6); d.append(ctx.stream_elapsed() * 1e3 if marks else 0.0)
med = lambda v: sorted(v)[len(v) // 2]
print("n = %d (%s): queueing %.1f us, wall to completion %.1f us (%.2f per generation), device between marks %.1f us (%.2f per generation)"
      % (n, os.environ.get("LMCMA_HIP_GRAPH_GENS", "default"), med(q), med(w), med(w) / n, med(d), med(d) / n))
