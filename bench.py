#!/usr/bin/env python3
"""Headline benchmark: candidate-trajectories/sec of the LM-CMA generation loop
(sample -> trajectory cost -> rank/select -> recombine -> update) on the 512^2
grid / 64-waypoint / lambda=4096 / m=16 configuration, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started without WORLD_SIZE and with --gpus N > 1 the program launches its own N ranks
(a child torch.distributed.run; the parent never touches a GPU) and refuses when the
node has fewer GPUs than ranks (--oversubscribe lets ranks share GPUs: a rehearsal).

A step is one generation over the whole population.  With N > 1 the population
grows with N (4096 candidates per GPU: weak scaling) and the shards exchange twice per
generation: their sorted costs and one row of N + 2 doubles each (elite partial sum,
success-rule count, lowest cost) -- directly into each other's mailboxes over xGMI
(--exchange peer, the default; checked against the RCCL form bit for bit before the
timed region, which falls back to RCCL if the check fails) or by two RCCL all-gathers
(--exchange library); the other form is timed as well and reported beside the headline.  The optimiser is put
back to its initial state every EPISODE generations (a new planning query on the
same map), so that every timed generation works on a live search distribution
instead of one that has already collapsed (sigma < 1e-20 is the reference's stop
condition, lmcma.cpp:426-429, reached after a few hundred generations here).

Rank 0 prints one JSON line; besides the throughput it carries
  roofline      one generation's algorithmic bytes (bytes per candidate-trajectory x
                candidates per GPU) over its device time (HIP events on the library's
                stream around the timed steps), against the 8 TB/s HBM3E peak
  cpu_baseline  the CPU oracle (a port of the reference algorithm, fp64) on a bounded
                sample of the same workload, candidates spread over the box's host
                threads; cpu_baseline_1thread: the same on one thread;
                cpu_baseline_reference: the reference's own compiled optimiser on one
                thread (where oracle/_ref was built).  N = 1 only
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
PRE_GENERATIONS = 200  # untimed, before the caller's --warmup
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
    if kernel == "select":      # read every cost and last generation's sorted list; write sorted costs and
        # ranks; the ranking launch also reads the elite rows it recombines (half of the local rows on
        # average, mu = lambda / 2) and writes one fp64 partial row per 32 candidates
        return lam_total * (4 + 4 + 8) + lam_local * 2 * N + (lam_local // 32) * 8 * N
    if kernel == "recombine":   # row sums: the partial rows in, one row out
        return (lam_local // 32) * 8 * N + 8 * N
    if kernel == "update":      # direction store in and out, the partial rows, the float images out
        return 8 * (4 * m * N) + min(lam_local // 32, 192) * 8 * N + 8 * m * N
    raise KeyError(kernel)


def cpu_baseline(prob, threads, budget_s):
    """The oracle's generation loop (the reference's ask/tell loop in C: Philox z, fp64
    sample/update, fp64 cost of the float-rounded rows) on `threads` host threads for about
    budget_s seconds.  Candidates are spread over the threads for sampling and scoring; the
    update stays serial as in the reference."""
    from oracle import pyoracle
    pyoracle.set_threads(threads)
    co = pyoracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
    o = pyoracle.OracleLMCMA(prob.N, prob.lam, prob.m, initial=prob.x0, lo=prob.lo, hi=prob.hi,
                             sigma=prob.sigma0, seed=42, z_source=pyoracle.Z_PHILOX)
    o.run_trajectory(co, 1)     # thread start-up, page faults
    gens, chunk = 0, 2
    t0 = time.perf_counter()
    while True:
        o.run_trajectory(co, chunk)
        gens += chunk
        el = time.perf_counter() - t0
        if el >= budget_s or gens >= 4000:
            break
        chunk = max(2, min(64, int(chunk * budget_s / max(el, 1e-3) / 8)))
    pyoracle.set_threads(1)
    return {"value": prob.lam * gens / el, "unit": "candidate-trajectories/s", "cores": threads, "kind": "port",
            "sample": "%d generations of lambda=%d (%.1f s) of the same workload, fp64, %d host thread%s"
                      % (gens, prob.lam, el, threads, "" if threads == 1 else "s over the candidates")}


def cpu_reference(prob, budget_s):
    """The reference's own optimiser (its lmcma.cpp compiled where it lies into oracle/_ref, see
    oracle/Makefile) driving its per-candidate ask/tell loop on one thread, with the oracle's fp64
    trajectory cost as the caller's cost function.  The reference keeps as many direction vectors as
    candidates (lmcma.cpp:266) and has no other mode, so what is timed is the first 32 generations of
    fresh runs at this workload's shape, in which its store holds 1 .. 32 vectors (16.5 on average,
    against the 16 of the configuration named); its own sequential generator draws the normals.
    None when the library was not built (it exists only where the reference tree was present at
    build time)."""
    from oracle import pyoracle, pyref
    if not pyref.available():
        return None
    import ctypes
    co = pyoracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
    cb = ctypes.cast(pyoracle.load().orc_cost_callback, ctypes.c_void_p)
    user = ctypes.cast(ctypes.byref(co.p), ctypes.c_void_p)
    gens_per_run, runs, seconds = 32, 0, 0.0
    while seconds < budget_s and runs < 64:
        r = pyref.RefLMCMA(prob.N, prob.lam, initial=prob.x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=42 + runs)
        seconds += r.run_timed(cb, user, gens_per_run)
        runs += 1
    return {"value": prob.lam * gens_per_run * runs / seconds, "unit": "candidate-trajectories/s", "cores": 1,
            "kind": "reference",
            "sample": "%d fresh runs of the reference's LMCMA (m = lambda = %d, its only mode), the first %d generations "
                      "each (1 to %d stored direction vectors), %.1f s, one thread, its own normal generator, fp64 cost"
                      % (runs, prob.lam, gens_per_run, gens_per_run, seconds)}


def batched_leg(hip, prob, lam, queries, steps):
    """The same workload with `queries` independent optimisers (restarts with their own seeds) in
    one context: every launch advances all of them.  Not the headline: reported beside it."""
    ctx = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start,
                         goal=prob.goal, w_len=prob.weights[0], w_grid=prob.weights[1], w_smooth=prob.weights[2],
                         queries=queries)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)

    def run_steps(k):
        done = 0
        while done < k:
            n = min(EPISODE, k - done)
            ctx.opt_reset()
            ctx.run(n)
            done += n

    run_steps(12)
    ctx.synchronize()
    run_steps(100)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.stream_mark(0)
    run_steps(steps)
    ctx.stream_mark(1)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    device_ms = ctx.stream_elapsed()
    ctx.close()
    b_cand = prob.algorithmic_bytes_per_candidate(lam)
    value = queries * lam * steps / elapsed
    achieved = b_cand * lam * queries / (device_ms * 1e-3 / steps) / 1e9
    return {"queries_in_flight": queries, "value": value, "unit": "candidate-trajectories/s", "steps": steps,
            "ms_per_step": elapsed / steps * 1e3, "us_per_generation_per_query": device_ms * 1e3 / steps / queries,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS},
            "note": "independent restarts of the same planning problem (own Philox keys) advanced by every launch"}


def host_threads():
    """Host threads of this job's CPU share: OMP_NUM_THREADS when the box sets it, else the
    affinity mask, at most 16."""
    try:
        n = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        n = 0
    if n < 1:
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))


class stdout_to_stderr:
    """RCCL writes a version banner to file descriptor 1 when its first communicator comes up;
    the one line this program owes its caller on stdout is the JSON record, so communicators are
    made with descriptor 1 pointing at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=2, help="benchmark configuration (2 = the headline one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--queries", type=int, default=1,
                    help="independent optimisers (restarts of the planning problem, own seeds) advanced together by "
                    "every launch of one context; 1 = the configuration as named")
    ap.add_argument("--batched-queries", type=int, default=64,
                    help="after the headline measurement, also measure this many queries in one context (0: skip)")
    ap.add_argument("--structured-covariance", action="store_true",
                    help="sample with the smoothness covariance of the reference (z <- L z on the matrix cores); "
                    "not the headline workload")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL) for real runs; gloo only to rehearse the "
                    "N > 1 code path on a box with fewer GPUs than ranks")
    ap.add_argument("--sharded-loop", action="store_true",
                    help="N = 1 only: go through lmcma_hip_run_sharded with a one-rank RCCL communicator, i.e. "
                    "what every rank of an N > 1 job runs (not the headline: the graph-replayed lmcma_hip_run is)")
    ap.add_argument("--exchange", default="peer", choices=["peer", "library", "torch", "rows"],
                    help="N > 1: 'peer' = the sharded loop inside the library with the direct exchange (every shard "
                    "stores its blocks into the peers' mailboxes, the consuming kernels wait for arrival counters); "
                    "'library' = the same loop with two RCCL all-gathers per generation; 'torch' = the loop driven "
                    "from Python with torch.distributed all-gathers (a rehearsal form); 'rows' = the rows exchange: every "
                    "rank holds the whole population, samples its range, and RCCL all-gathers rows and costs (the state "
                    "is one context's bit for bit whatever N; the slow, G-independent form)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="let the ranks share the GPUs when the node has fewer than --gpus (a rehearsal of the N > 1 "
                    "path, not a measurement: backend gloo, the direct exchange over IPC on one device)")
    args = ap.parse_args()

    import torch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # This process starts the ranks and never touches a GPU itself (counting devices does not).
        have = torch.cuda.device_count()
        if have < args.gpus and not args.oversubscribe:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); refusing to run fewer ranks than asked "
                             "for (--oversubscribe rehearses the path with ranks sharing GPUs)" % (args.gpus, have))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch.distributed as dist
    from lmcma_path_planner_amd import grids, hip
    from lmcma_path_planner_amd import dist as shard_mod

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > torch.cuda.device_count():
        if not args.oversubscribe:
            raise SystemExit("%d ranks on %d GPU(s): --oversubscribe is needed for that" % (world, torch.cuda.device_count()))
        args.backend = "gloo"       # RCCL refuses two ranks on one device
        if args.exchange in ("library", "rows"):
            raise SystemExit("--oversubscribe: the RCCL exchanges cannot run with ranks sharing a GPU")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(args.backend)

    prob = grids.benchmark_problem(args.config)
    lam_local = prob.lam if args.config <= 3 else prob.lam // 8   # cfg4/5 are quoted as 8-GPU populations
    lam_total = lam_local * world
    native = world > 1 and args.exchange in ("peer", "library", "rows")
    have_rccl = world > 1 and args.backend == "nccl"
    if world > 1 and args.exchange in ("library", "rows") and not have_rccl:
        raise SystemExit("--exchange %s needs --backend nccl" % args.exchange)
    rows_form = world > 1 and args.exchange == "rows"
    stream = None
    if world > 1 and not native:
        # kernels and collectives are ordered on one non-default torch stream
        tstream = torch.cuda.Stream(device)
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    ctx = hip.HipContext(prob.N, lam_total, prob.m, dims=prob.dims, waypoints=prob.W,
                         lambda_local=lam_total if rows_form else lam_local,
                         first_candidate=0 if rows_form else rank * lam_local, seed=42, start=prob.start, goal=prob.goal,
                         w_len=prob.weights[0], w_grid=prob.weights[1], w_smooth=prob.weights[2],
                         device=local_rank, stream=stream, queries=args.queries)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    if args.structured_covariance:
        from lmcma_path_planner_amd import planner
        ctx.set_cov_factor(planner.cholesky(planner.covariance(prob.dims, prob.W)))
    ctx.opt_init(prob.x0, prob.sigma0 * (8.0 if args.structured_covariance else 1.0))

    # An episode (one planning query: EPISODE generations from the initial state) runs on across
    # the calls of run_steps: warm-up and timed region are stretches of the same sequence of
    # episodes, and a reset falls where an episode ends, not where a region starts.
    episode = [0]
    exchange_used, exchange_check = ("torch" if world > 1 else "none"), None
    ctx_peer_mapped = [False]   # the mailboxes are mapped and the direct form gave the RCCL form's bits
    if world == 1 and args.sharded_loop:
        with stdout_to_stderr():
            ctx.comm_init(hip.HipContext.comm_unique_id(), 1, 0)

        def run_steps(k):
            done = 0
            while done < k:
                n = min(EPISODE - episode[0], k - done)
                if episode[0] == 0:
                    ctx.opt_reset()
                ctx.run_sharded(n)
                done += n
                episode[0] = (episode[0] + n) % EPISODE
    elif world == 1:
        def run_steps(k):
            done = 0
            while done < k:
                n = min(EPISODE - episode[0], k - done)
                if episode[0] == 0:
                    ctx.opt_reset()
                ctx.run(n)
                done += n
                episode[0] = (episode[0] + n) % EPISODE
    elif native:
        with stdout_to_stderr():
            if have_rccl:
                # one RCCL communicator of the library's own: rank 0 makes the id, torch.distributed carries it
                uid = [hip.HipContext.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                ctx.comm_init(uid[0], world, rank)
            if args.exchange == "peer":
                # the mailboxes: every shard's handle to every shard, all mapped before anyone sends
                handles = [None] * world
                problem = None
                try:
                    mine = ctx.peer_handle()
                except Exception as e:      # (the library's error text says what was refused)
                    mine, problem = None, str(e)
                dist.all_gather_object(handles, mine)
                if problem is None and all(h is not None for h in handles):
                    try:
                        ctx.peer_connect(handles, rank)
                    except Exception as e:
                        problem = str(e)
                elif problem is None:
                    problem = "another shard could not export its mailbox"
                problems = [None] * world
                dist.all_gather_object(problems, problem)
                problems = [p for p in problems if p]
                exchange_used = "peer"
                if problems:
                    # every shard or none: back to the communicator, or stop where there is none
                    try:
                        ctx.peer_disconnect()
                    except Exception:
                        pass
                    if not have_rccl:
                        raise SystemExit("the mailboxes of the direct exchange could not be set up: %s" % problems[0])
                    exchange_used = "library"
                    exchange_check = "the mailboxes could not be set up (%s): the RCCL form is what was timed" % problems[0]
                dist.barrier()
                if problems:
                    pass
                elif have_rccl:
                    # the direct exchange against the RCCL form on this very node: same bits, or RCCL it is
                    def probe():
                        ctx.opt_reset()
                        ctx.run_sharded(6)
                        ctx.synchronize()
                        try:
                            st_ = ctx.opt_get()
                        except Exception:
                            return None
                        return np.concatenate([st_["xmean"], [st_["sigma"], st_["s"], st_["best_f"]]]).tobytes()
                    a_ = probe()
                    ctx.peer_enable(False)
                    b_ = probe()
                    okf = torch.tensor([1 if (a_ is not None and a_ == b_) else 0], device=device)
                    dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                    exchange_check = "identical to the RCCL form over 6 generations" if int(okf.item()) == 1 else \
                        "DIFFERS from the RCCL form or timed out: the RCCL form is what was timed"
                    if int(okf.item()) == 1:
                        # both forms give the same bits here: the faster one on this node is the one timed
                        # (the other is timed beside it afterwards, exchange_alternative)
                        def rate(peer_on, episodes=3):
                            ctx.peer_enable(peer_on)
                            ctx.opt_reset()
                            ctx.run_sharded(40)
                            ctx.synchronize()
                            dist.barrier()
                            t0_ = time.perf_counter()
                            for _ in range(episodes):       # (episodes as in the timed region: no run into the stop rule)
                                ctx.opt_reset()
                                ctx.run_sharded(EPISODE)
                            ctx.synchronize()
                            t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=device)
                            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                            return float(t_.item()) / (episodes * EPISODE) * 1e6
                        ctx_peer_mapped[0] = True
                        us_peer, us_rccl = rate(True), rate(False)
                        use_peer = us_peer <= us_rccl        # (the same figures on every rank: the same choice)
                        ctx.peer_enable(use_peer)
                        exchange_used = "peer" if use_peer else "library"
                        exchange_check += "; %.1f us per generation with the direct exchange, %.1f with RCCL over 300 " \
                                          "generations: the faster one is what was timed" % (us_peer, us_rccl)
                    else:
                        exchange_used = "library"
                    ctx.opt_reset()
                else:
                    exchange_check = "not checked (no RCCL with ranks sharing a GPU)"
            else:
                exchange_used, exchange_check = args.exchange, None

        def run_steps(k):
            done = 0
            while done < k:
                n = min(EPISODE - episode[0], k - done)
                if episode[0] == 0:
                    ctx.opt_reset()
                ctx.run_sharded(n)
                done += n
                episode[0] = (episode[0] + n) % EPISODE
    else:
        shard = shard_mod.HipShard(ctx, device)

        def run_steps(k):
            for g in range(k):
                if episode[0] == 0:
                    ctx.opt_reset()
                shard_mod.sharded_generation(shard)
                episode[0] = (episode[0] + 1) % EPISODE

    def fence():
        # the library's own work first: its collectives (a communicator of its own, on its own
        # stream) have finished on this rank before torch.distributed's communicator runs the barrier
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Outside every timed region, whatever --warmup is: the library records its HIP graphs on the
    # first run call (one eager generation, then the captures), modules load, LDS limits are raised,
    # and PRE_GENERATIONS of the same workload (a few milliseconds) bring the clocks up from idle.
    with stdout_to_stderr():    # (torch.distributed brings its communicator up at the first barrier)
        run_steps(PRE_GENERATIONS)
        fence()
        run_steps(args.warmup)
        fence()
        if args.steps <= 256:
            # A short timed region is one library call.  Like any caller that steps K generations at
            # a time it has made that call before: the library keeps a graph of a call length it has
            # seen, recorded behind the first such call and replayed from the second on.  A graph's
            # first launches cost more than its later ones (tools/short_call.py, 20 generations: 526,
            # 508, 504 us ... 496 from the seventh on), hence three rehearsals: the call behind which
            # the graph is recorded and two replays.  (With --steps 20 --warmup 5 the
            # timed call then covers generations 65 to 85 of an episode; a fourth rehearsal would put
            # the episode's end, one reset and a call cut in two, into a region of 20 generations.)
            for _ in range(3):
                run_steps(args.steps)
                fence()
    t0 = time.perf_counter()
    ctx.stream_mark(0)
    run_steps(args.steps)
    ctx.stream_mark(1)
    fence()
    elapsed = time.perf_counter() - t0
    device_ms = ctx.stream_elapsed()        # HIP events on the library's stream around the same K steps
    rdev = device if args.backend == "nccl" else torch.device("cpu")   # where the ranks' scalars are reduced
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-kernel launch durations: a second, eager pass with an event bracket around every launch.
    # A bracket reads more than the kernel inside it.  The kernels of a generation run back to back
    # on one stream, so their durations add up to the generation's device time, which the events
    # around the replayed steps above give (device_ms, not the wall clock): the share of a bracket
    # that is not kernel is (sum of the brackets per generation - that time) / brackets per generation.
    # The rocprofv3 averages under profiles/ are the reference for these durations.
    timed_steps = min(args.steps, 400)
    ctx.timing_enable(True)
    ctx.stream_mark(0)
    run_steps(timed_steps)
    ctx.stream_mark(1)
    fence()
    eager_ms = ctx.stream_elapsed()
    timing = ctx.timing_get()
    ctx.timing_enable(False)
    empty_bracket_us = ctx.timing_event_overhead(1024) * 1e3
    brackets = sum(n for (_, n) in timing.values())
    bracket_sum_us = sum(ms for (ms, _) in timing.values()) * 1e3
    gen_device_us = device_ms * 1e3 / args.steps
    event_overhead_us = (max((bracket_sum_us / timed_steps - gen_device_us) / max(brackets / timed_steps, 1), 0.0)
                         if world == 1 else 0.0)   # collectives sit between a shard's kernels: brackets as read
    st = ctx.opt_get()

    if rank == 0:
        per_kernel = {}
        for name, (ms, n) in timing.items():
            if n:
                avg_s = max(ms * 1e-3 / n - event_overhead_us * 1e-6, 1e-7)
                b = algorithmic_bytes_per_launch(name, prob, lam_local, lam_total)
                per_kernel[name] = {"avg_us": avg_s * 1e6, "bracket_us": ms * 1e3 / n, "launches": n, "bytes": b,
                                    "GBps": b / avg_s / 1e9}
        if "update" in per_kernel and "generate" in per_kernel and per_kernel["update"]["launches"] < per_kernel["generate"]["launches"]:
            # between two generations of one call the update runs at the head of the next sampling launch
            g_ = per_kernel["generate"]
            g_["carries"] = "the second part of the previous generation's update (all but the last generation of a call)"
            g_["bytes"] += algorithmic_bytes_per_launch("update", prob, lam_local, lam_total)
            g_["GBps"] = g_["bytes"] / (g_["avg_us"] * 1e-6) / 1e9
        dominant = max(per_kernel, key=lambda k: per_kernel[k]["avg_us"] * per_kernel[k]["launches"])
        # One generation = the chain of launches the library replays from its graph; its algorithmic
        # bytes are bytes-per-candidate x the candidates this GPU owns, its duration the device time of
        # the timed region / K (events on the library's stream).  That chain is what the roofline
        # fraction is quoted for: no single kernel of it moves the generation's bytes.
        b_cand = prob.algorithmic_bytes_per_candidate(lam_local)
        gen_bytes = b_cand * lam_local * args.queries
        gen_us = device_ms * 1e3 / args.steps
        achieved = gen_bytes / (gen_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_cfg%d.json" % args.config)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get("generation")
        out = {
            "metric": "candidate-trajectories/sec (lambda x gens/s), 512^2 grid, 64-waypoint",
            "value": args.queries * lam_total * args.steps / elapsed,
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
                       "lambda_total": lam_total, "m": prob.m, "queries_in_flight": args.queries,
                       "z_source": "philox4x32-10",
                       "update": "on device, fp64", "episode_generations": EPISODE,
                       "exchange": "the candidates' rows and costs, two RCCL all-gathers in place; every rank ranks and "
                                   "recombines the whole population" if exchange_used == "rows" else
                                   ("sorted local costs + one row of N+2 doubles per shard (elite partial sum, "
                                    "success-rule count, lowest cost), " +
                                    {"peer": "stored by each shard into the peers' mailboxes (fine-grained memory mapped "
                                             "over IPC, arrival counters, the consuming kernels wait at their head)",
                                     "library": "two RCCL all-gathers queued by the library",
                                     "torch": "torch.distributed all-gathers from Python"}[exchange_used])
                       if world > 1 else "none",
                       "exchange_check": exchange_check},
            "roofline": {"bound": "hbm", "kernel": "one generation: " + " + ".join(k for k in per_kernel),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "bytes_per_launch": gen_bytes, "bytes_per_candidate": b_cand,
                         "avg_launch_us": gen_us,
                         "longest_kernel": dominant},
            "kernels": per_kernel,
            "event_brackets": {"overhead_us": event_overhead_us, "empty_bracket_us": empty_bracket_us,
                               "eager_pass_us_per_step": eager_ms * 1e3 / timed_steps},
            "final": {"generations": st["itr"], "best_f": st["best_f"], "sigma": st["sigma"]},
        }
        if world == 1 and args.queries == 1 and args.batched_queries > 1:
            out["batched_queries"] = batched_leg(hip, prob, lam_local, args.batched_queries, max(args.steps, 400))
        if world == 1 and not args.no_cpu_baseline:
            cores = host_threads()
            out["cpu_baseline"] = cpu_baseline(prob, cores, 12.0)
            out["cpu_baseline_1thread"] = cpu_baseline(prob, 1, 8.0)
            ref = cpu_reference(prob, 8.0)
            if ref is not None:
                out["cpu_baseline_reference"] = ref
    alt = None
    if world > 1 and native and have_rccl and args.exchange == "peer" and ctx_peer_mapped[0]:
        # the other exchange form on the same contexts, timed the same way (a shorter region)
        other_is_peer = exchange_used != "peer"
        ctx.peer_enable(other_is_peer)
        k_alt = min(args.steps, 400)
        run_steps(100)
        fence()
        t0 = time.perf_counter()
        run_steps(k_alt)
        fence()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        alt = {"exchange": "peer" if other_is_peer else "library (two RCCL all-gathers)",
               "value": lam_total * k_alt / float(t.item()), "unit": "candidate-trajectories/s", "steps": k_alt,
               "ms_per_step": float(t.item()) / k_alt * 1e3}
    if rank == 0:
        if alt is not None:
            out["exchange_alternative"] = alt
        if world > 1 and exchange_used == "peer":
            out["peer_status"] = ctx.peer_status()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()      # nobody unmaps a mailbox a peer may still write into
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
