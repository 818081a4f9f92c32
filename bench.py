#!/usr/bin/env python3
"""Headline benchmark: candidate-trajectories/sec of the LM-CMA generation loop
(sample -> trajectory cost -> rank/select -> recombine -> update) on the 512^2
grid / 64-waypoint / lambda=4096 / m=16 configuration, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one generation over the whole population.  With N > 1 the population
grows with N (4096 candidates per GPU: weak scaling) and two all-gathers per
generation exchange the costs and the elite partial sums.  The optimiser is put
back to its initial state every EPISODE generations (a new planning query on the
same map), so that every timed generation works on a live search distribution
instead of one that has already collapsed (sigma < 1e-20 is the reference's stop
condition, lmcma.cpp:426-429, reached after a few hundred generations here).

Rank 0 prints one JSON line; besides the throughput it carries
  roofline      the dominant kernel's algorithmic bytes per launch over its
                HIP-event launch duration, against the 8 TB/s HBM3E peak
  cpu_baseline  the CPU oracle (a port of the reference algorithm, fp64, one
                thread) on a bounded sample of the same workload, N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
EPISODE = 100          # generations per planning query


def algorithmic_bytes_per_launch(kernel, prob, lam_local, lam_total):
    """HBM bytes one launch of each kernel class has to move if every operand is
    read or written exactly once (fp32 rows, fp32 costs, int32 ranks)."""
    N, m = prob.N, prob.m
    if kernel == "sample":      # write the rows; read V and B once
        return lam_local * 4 * N + 8 * m * N
    if kernel == "cost":        # read the rows and the grid once; write one cost
        return lam_local * (4 * N + 4) + prob.grid_bytes()
    if kernel == "generate":    # fused: write the rows, read V, B and the grid once, write one cost
        return lam_local * (4 * N + 4) + 8 * m * N + prob.grid_bytes()
    if kernel == "select":      # read every cost; write sorted costs and ranks
        return lam_total * (4 + 8)
    if kernel == "recombine":   # read the local half of the mu = lambda/2 elite rows
        return lam_local * 2 * N + (lam_total // 2) * 12
    if kernel == "update":      # direction store in and out, previous/current sorted costs
        return 8 * (4 * m * N) + 8 * lam_total
    raise KeyError(kernel)


def cpu_baseline(prob, budget_s=12.0):
    """The oracle's generation loop (Philox z, fp64 sample/update, fp64 cost on
    fp32-rounded rows) on the host, single thread, for about budget_s seconds."""
    from oracle import pyoracle
    co = pyoracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
    o = pyoracle.OracleLMCMA(prob.N, prob.lam, prob.m, initial=prob.x0, lo=prob.lo, hi=prob.hi,
                             sigma=prob.sigma0, seed=42, z_source=pyoracle.Z_PHILOX)
    gens = 0
    t0 = time.perf_counter()
    while True:
        c, _ = co.cost_rows_f32(o.arx.astype(np.float32))
        o.tell_all(c)
        gens += 1
        el = time.perf_counter() - t0
        if el >= budget_s or gens >= 2000:
            break
    return {"value": prob.lam * gens / el, "unit": "candidate-trajectories/s", "cores": 1, "kind": "port",
            "sample": "%d generations of lambda=%d (%.1f s) of the same workload, fp64, one thread" % (gens, prob.lam, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=2, help="benchmark configuration (2 = the headline one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--structured-covariance", action="store_true",
                    help="sample with the smoothness covariance of the reference (z <- L z on the matrix cores); "
                    "not the headline workload")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL) for real runs; gloo only to rehearse the "
                    "N > 1 code path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lmcma_path_planner_amd import grids, hip
    from lmcma_path_planner_amd import dist as shard_mod

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    prob = grids.benchmark_problem(args.config)
    lam_local = prob.lam if args.config <= 3 else prob.lam // 8   # cfg4/5 are quoted as 8-GPU populations
    lam_total = lam_local * world
    stream = None
    if world > 1:
        # kernels and collectives are ordered on one non-default torch stream
        tstream = torch.cuda.Stream(device)
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    ctx = hip.HipContext(prob.N, lam_total, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=lam_local,
                         first_candidate=rank * lam_local, seed=42, start=prob.start, goal=prob.goal,
                         w_len=prob.weights[0], w_grid=prob.weights[1], w_smooth=prob.weights[2],
                         device=local_rank, stream=stream)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    if args.structured_covariance:
        from lmcma_path_planner_amd import planner
        ctx.set_cov_factor(planner.cholesky(planner.covariance(prob.dims, prob.W)))
    ctx.opt_init(prob.x0, prob.sigma0 * (8.0 if args.structured_covariance else 1.0))

    if world == 1:
        def run_steps(k):
            done = 0
            while done < k:
                n = min(EPISODE, k - done)
                ctx.opt_reset()
                ctx.run(n)
                done += n
    else:
        shard = shard_mod.HipShard(ctx, device)

        def run_steps(k):
            for g in range(k):
                if g % EPISODE == 0:
                    ctx.opt_reset()
                shard_mod.sharded_generation(shard)

    def fence():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations with HIP events on the library's stream, same number of steps
    ctx.timing_enable(True)
    if world == 1:
        run_steps(args.steps)
    else:
        run_steps(min(args.steps, 200))
    fence()
    timing = ctx.timing_get()
    ctx.timing_enable(False)
    # An event bracket also reads time between the two records that is not the kernel (a few
    # microseconds on this stack).  On one GPU the kernels of a generation run back to back
    # on one stream, so their durations add up to the measured time per generation: the
    # share of a bracket that is not kernel is chosen to make them do so.  (An empty bracket
    # reads more than that share, a bracket around an empty kernel less; both are printed.)
    empty_bracket_us = ctx.timing_event_overhead(1024) * 1e3
    timed_steps = args.steps if world == 1 else min(args.steps, 200)
    brackets_per_step = sum(n for (_, n) in timing.values()) / timed_steps
    bracket_us_per_step = sum(ms for (ms, _) in timing.values()) * 1e3 / timed_steps
    if world == 1:
        event_overhead_us = max((bracket_us_per_step - elapsed / args.steps * 1e6) / brackets_per_step, 0.0)
    else:
        event_overhead_us = 0.0   # collectives sit between the kernels: the brackets are reported as read
    st = ctx.opt_get()

    if rank == 0:
        per_kernel = {}
        for name, (ms, n) in timing.items():
            if n:
                avg_s = max(ms * 1e-3 / n - event_overhead_us * 1e-6, 1e-7)
                b = algorithmic_bytes_per_launch(name, prob, lam_local, lam_total)
                per_kernel[name] = {"avg_us": avg_s * 1e6, "bracket_us": ms * 1e3 / n, "launches": n, "bytes": b,
                                    "GBps": b / avg_s / 1e9}
        dominant = max(per_kernel, key=lambda k: per_kernel[k]["avg_us"] * per_kernel[k]["launches"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_cfg%d.json" % args.config)
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dominant)
        achieved = per_kernel[dominant]["GBps"]
        out = {
            "metric": "candidate-trajectories/sec (lambda x gens/s), 512^2 grid, 64-waypoint",
            "value": lam_total * args.steps / elapsed,
            "unit": "candidate-trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": prob.name + ("+structured-covariance" if args.structured_covariance else ""), "grid": "x".join(str(s) for s in prob.cells.shape),
                       "waypoints": prob.W, "n_params": prob.N, "lambda_per_gpu": lam_local,
                       "lambda_total": lam_total, "m": prob.m, "z_source": "philox4x32-10",
                       "update": "on device, fp64", "episode_generations": EPISODE, "exchange": "all-gather of sorted local costs + all-gather of elite partial sums (N+2 doubles per rank)"
                       if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": per_kernel[dominant]["bytes"],
                         "avg_launch_us": per_kernel[dominant]["avg_us"],
                         "event_bracket_us": per_kernel[dominant]["bracket_us"],
                         # the whole generation against the same peak: algorithmic bytes per
                         # candidate-trajectory x candidates per second (all ranks)
                         "generation_achieved": lam_total * args.steps / elapsed *
                         prob.algorithmic_bytes_per_candidate(lam_local) / 1e9,
                         "generation_frac": lam_total * args.steps / elapsed *
                         prob.algorithmic_bytes_per_candidate(lam_local) / 1e9 / (HBM_PEAK_GBS * world),
                         "bracket_overhead_us": event_overhead_us,
                         "empty_bracket_us": empty_bracket_us},
            "kernels": per_kernel,
            "generation_bytes_per_candidate": prob.algorithmic_bytes_per_candidate(lam_local),
            "final": {"generations": st["itr"], "best_f": st["best_f"], "sigma": st["sigma"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
