"""Time the per-generation phases one rank of a G-GPU job runs, on a single GPU.

    python tools/shard_phases.py <G> [generations]

The rank owns the first 4096 candidates of a population of 4096 G (cfg2 shapes, weak
scaling as in bench.py); the costs of the other ranks' candidates are filled once with
plausible values and the G partial sums with copies of the rank's own, so the kernels see
the sizes of the real job.  Run it under `rocprofv3 --kernel-trace --stats` for per-kernel
times; by itself it prints the wall time per generation without the collectives."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from lmcma_path_planner_amd import grids, hip, dist as shard_mod

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prob = grids.benchmark_problem(2)
lam_local, lam_total = prob.lam, prob.lam * G
device = torch.device("cuda", 0)
tstream = torch.cuda.Stream(device)
torch.cuda.set_stream(tstream)
ctx = hip.HipContext(prob.N, lam_total, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=lam_local,
                     first_candidate=0, seed=42, start=prob.start, goal=prob.goal, stream=tstream.cuda_stream)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
shard = shard_mod.HipShard(ctx, device)
rng = np.random.default_rng(0)
# the other contexts' blocks: ascending costs, as they arrive from the exchange
other = np.sort(rng.gamma(2.0, 20000.0, size=(G, lam_local)).astype(np.float32), axis=1)
shard.cost_all.copy_(torch.from_numpy(other.reshape(-1)))


def generation():
    shard.phase_local()
    shard.phase_rank()
    if G > 1:
        L = shard.partial_len
        pa = shard.partial_all[: G * L].view(G, L)
        pa.copy_(shard.partial_local.expand(G, L))
        pa[:, : prob.N] /= G
        shard.phase_update(G)
    else:
        shard.phase_update(0)


for g in range(20):
    generation()
ctx.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for g in range(gens):
    if g % 100 == 0:
        ctx.opt_reset()
    generation()
ctx.synchronize(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / gens
if "diag" in os.environ.get("LMCMA_HIP_LIB", ""):
    import ctypes
    lib = ctx.lib
    lib.lmcma_hip_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    sr = (ctypes.c_uint64 * 16)()
    # the merge kernel is the last ranking launch of a generation: its stamps are the ones left
    shard.phase_local(); ctx.synchronize(); shard.phase_rank(); ctx.synchronize()
    lib.lmcma_hip_diag_stamps(ctx._h, 1, sr)
    names = ["own key + pivots", "pivot searches + windows", "publish", "elite rows + stores"]
    print("merge kernel (last workgroup), ns:", {names[i]: int(sr[9 + i] - sr[8 + i]) * 10 for i in range(4)})
print("G = %d: lambda_total %d, %.1f us per generation without collectives (eager launches)" % (G, lam_total, dt * 1e6))
