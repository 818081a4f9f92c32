"""Host time per sharded generation: the eager loop of bench.py --gpus N, on one GPU.

    python tools/shard_host_overhead.py [generations]

A one-rank RCCL group stands in for the node: the collectives move nothing but go through
the same torch / RCCL launch path, and the kernels see the buffer sizes of a 2-rank job.
Prints the wall time per generation of (a) the library phases alone, (b) phases +
collectives as bench.py --exchange torch issues them, (c) the host time to enqueue (b) without
waiting, and (d) the same for the loop inside the library (lmcma_hip_run_sharded with a one-rank
communicator of its own: phases and RCCL calls queued from C++)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import torch.distributed as dist
from lmcma_path_planner_amd import grids, hip, dist as shard_mod

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 300
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
G = 2
prob = grids.benchmark_problem(2)
tstream = torch.cuda.Stream(device)
torch.cuda.set_stream(tstream)
ctx = hip.HipContext(prob.N, prob.lam * G, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=prob.lam,
                     first_candidate=0, seed=42, start=prob.start, goal=prob.goal, stream=tstream.cuda_stream)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
shard = shard_mod.HipShard(ctx, device)
rng = np.random.default_rng(0)
shard.cost_all.copy_(torch.from_numpy(np.sort(rng.gamma(2.0, 20000.0, size=(G, prob.lam)).astype(np.float32), axis=1).reshape(-1)))
L = shard.partial_len
scratch_c = torch.empty_like(shard.cost_local)
scratch_p = torch.empty(L, dtype=torch.float64, device=device)


def phases_only():
    shard.phase_local(); shard.phase_rank()
    shard.partial_all[L: 2 * L].copy_(shard.partial_local)
    shard.phase_update(G)


def with_collectives():
    shard.phase_local()
    dist.all_gather_into_tensor(scratch_c, shard.cost_local)     # one rank: same launch path, no data moved
    shard.phase_rank()
    dist.all_gather_into_tensor(scratch_p, shard.partial_local)
    shard.partial_all[L: 2 * L].copy_(shard.partial_local)
    shard.phase_update(G)


def timed(fn, wait=True):
    for g in range(20):
        fn()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in range(gens):
        if g % 100 == 0:
            ctx.opt_reset()
        fn()
    t_enq = time.perf_counter() - t0
    ctx.synchronize(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return t_enq / gens * 1e6, t_all / gens * 1e6


e, a = timed(phases_only)
print("phases only:        enqueue %.1f us, complete %.1f us per generation" % (e, a))
e, a = timed(with_collectives)
print("with 2 collectives: enqueue %.1f us, complete %.1f us per generation" % (e, a))
ctx.close()

# (d) the loop inside the library
nat = hip.HipContext(prob.N, prob.lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start,
                     goal=prob.goal)
nat.set_grid(prob.cells); nat.set_bounds(prob.lo, prob.hi); nat.opt_init(prob.x0, prob.sigma0)
nat.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
nat.run_sharded(20)
nat.synchronize()
t0 = time.perf_counter()
for g0 in range(0, gens, 100):
    nat.opt_reset()
    nat.run_sharded(min(100, gens - g0))
t_enq = time.perf_counter() - t0
nat.synchronize()
t_all = time.perf_counter() - t0
print("inside the library: enqueue %.1f us, complete %.1f us per generation" % (t_enq / gens * 1e6, t_all / gens * 1e6))
nat.close()
dist.destroy_process_group()
