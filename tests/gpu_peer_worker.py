"""Worker of test_direct_exchange_over_processes: one shard of a population, one process per shard,
every process on GPU 0 of the test box (the shards' mailboxes are mapped through IPC handles).

Launched by torch.distributed.run; torch.distributed (gloo) only carries the handles and the final
comparison, the generations run through lmcma_hip_run_sharded with the direct exchange.
argv: benchmark problem, candidates per shard, generations."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from lmcma_path_planner_amd import grids, hip  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    cfg, nloc, gens = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    prob = grids.benchmark_problem(cfg)
    ctx = hip.HipContext(prob.N, world * nloc, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=nloc,
                         first_candidate=rank * nloc, start=prob.start, goal=prob.goal, seed=42, w_len=prob.weights[0],
                         w_grid=prob.weights[1], w_smooth=prob.weights[2], device=0)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)
    handles = [None] * world
    dist.all_gather_object(handles, ctx.peer_handle())
    ctx.peer_connect(handles, rank)
    dist.barrier()                      # every shard is connected before any shard sends
    for n in (1, gens // 2, gens - 1 - gens // 2):
        ctx.run_sharded(n)
    ctx.synchronize()
    status = ctx.peer_status()
    mine = dict(state=ctx.opt_get(), vectors=ctx.opt_vectors(), rows=ctx.rows(), costs=ctx.costs(), status=status)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    dist.barrier()                      # nobody unmaps a mailbox a peer may still write into
    ctx.close()
    ok = True
    if rank == 0:
        import test_gpu_sharded_multi as T
        assert all(e["status"]["wait_error"] == 0 for e in everyone), [e["status"] for e in everyone]
        want = T.by_phases(prob, world, nloc, gens)
        try:
            T.assert_same(everyone, want, gens)
            print("direct exchange over %d processes: identical (fine-grained mailbox: %s)"
                  % (world, everyone[0]["status"]["fine_grained"]), flush=True)
        except AssertionError as e:
            ok = False
            print("direct exchange over %d processes: DIFFERENT\n%s" % (world, e), flush=True)
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
