"""Worker of test_sharded_generation_over_two_processes: one rank of a two-rank job on one GPU.

Launched by torch.distributed.run with the gloo backend (two ranks cannot share one GPU under
RCCL); the ranks run lmcma_path_planner_amd.dist.sharded_generation on device buffers exactly as
bench.py does, and rank 0 compares the replicated state with an unsharded context."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmcma_path_planner_amd import dist as shard_mod  # noqa: E402
from lmcma_path_planner_amd import grids, hip  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    gens = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    prob = grids.benchmark_problem(2)
    nloc = 1024
    lam = nloc * world
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)

    def make(first, n_local, st):
        c = hip.HipContext(prob.N, lam, prob.m, dims=2, waypoints=prob.W, lambda_local=n_local, first_candidate=first,
                           start=prob.start, goal=prob.goal, seed=31, stream=st)
        c.set_grid(prob.cells)
        c.set_bounds(prob.lo, prob.hi)
        c.opt_init(prob.x0, prob.sigma0)
        return c

    ctx = make(rank * nloc, nloc, stream.cuda_stream)
    shard = shard_mod.HipShard(ctx, device)
    for _ in range(gens):
        shard_mod.sharded_generation(shard)
    ctx.synchronize()
    st = ctx.opt_get()
    # every rank holds the same replicated state
    mine = torch.from_numpy(np.concatenate([st["xmean"], [st["sigma"], st["best_f"]]]))
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = all(torch.equal(b, both[0]) for b in both)
    if rank == 0:
        whole = make(0, lam, None)
        whole.run(gens)
        w = whole.opt_get()
        ok = ok and st["itr"] == w["itr"] == gens
        ok = ok and np.allclose(st["xmean"], w["xmean"], rtol=1e-11, atol=1e-11)
        ok = ok and np.isclose(st["sigma"], w["sigma"], rtol=1e-11) and st["best_f"] == w["best_f"]
        whole.close()
        print("sharded vs whole: %s (sigma %.17g / %.17g)" % ("equal" if ok else "DIFFERENT", st["sigma"], w["sigma"]), flush=True)
    ctx.close()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
