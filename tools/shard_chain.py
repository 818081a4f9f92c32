"""One shard of a G-shard population timed on one GPU through lmcma_hip_run_sharded with the DIRECT
exchange: the other G - 1 shards are images of this one (its own mailbox handle in their places,
see lmcma_hip_peer_connect), so every launch, every push into a mailbox and every wait of a real
generation runs, with the arrival of the peers' blocks costing nothing (no link).

    python tools/shard_chain.py <G> [generations] [config] [candidates per shard]

Prints the device time per generation (events on the library's stream); under
`rocprofv3 --kernel-trace --stats` the per-kernel averages."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lmcma_path_planner_amd import grids, hip

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
prob = grids.benchmark_problem(cfg)
nloc = int(sys.argv[4]) if len(sys.argv) > 4 else (prob.lam if cfg <= 3 else prob.lam // 8)
rank = G // 2
ctx = hip.HipContext(prob.N, G * nloc, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=nloc, first_candidate=rank * nloc,
                     seed=42, start=prob.start, goal=prob.goal, w_len=prob.weights[0], w_grid=prob.weights[1],
                     w_smooth=prob.weights[2])
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
h = ctx.peer_handle()
ctx.peer_connect([h] * G, rank)


def run(k):
    done = 0
    while done < k:
        n = min(100, k - done)
        ctx.opt_reset()
        ctx.run_sharded(n)
        done += n


run(200)
ctx.synchronize()
t0 = time.perf_counter()
ctx.stream_mark(0)
run(gens)
ctx.stream_mark(1)
ctx.synchronize()
wall = time.perf_counter() - t0
dev = ctx.stream_elapsed()
st = ctx.peer_status()
print("G = %d, %d candidates per shard, cfg%d: %.2f us per generation on the device, %.2f us wall; waits given up: %d; "
      "%.3g candidate-trajectories/s for %d such shards" % (G, nloc, cfg, dev * 1e3 / gens, wall * 1e6 / gens, st["wait_error"],
                                                          G * nloc / (dev * 1e-3 / gens), G))
ctx.close()
