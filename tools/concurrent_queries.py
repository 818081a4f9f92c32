#!/usr/bin/env python3
"""Aggregate throughput of K independent planning queries on one map, each in its own context
(own stream, own HIP graphs), queued round-robin from one host thread.

    python tools/concurrent_queries.py [cfg] [generations per query] [K ...]
    python tools/concurrent_queries.py batch [cfg] [generations per query] [Q ...]

The second form puts the Q queries into ONE context (lmcma_hip_config.queries): every launch
advances all of them.

One query's generation is a chain of three dependent launches that leaves most of the chip
idle; this measures how much of that idle machine K concurrent queries can use.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batch_main(argv):
    cfg = int(argv[0]) if argv else 2
    gens = int(argv[1]) if len(argv) > 1 else 2000
    qs = [int(a) for a in argv[2:]] or [1, 2, 4, 8, 16, 32, 64]
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(cfg)
    lam = prob.lam if cfg <= 3 else prob.lam // 8
    for Q in qs:
        c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start,
                           goal=prob.goal, queries=Q)
        c.set_grid(prob.cells)
        c.set_bounds(prob.lo, prob.hi)
        c.opt_init(prob.x0, prob.sigma0)

        def run_all(n):
            done = 0
            while done < n:
                step = min(100, n - done)
                c.opt_reset()
                c.run(step)
                done += step

        run_all(112)
        c.synchronize()
        t0 = time.perf_counter()
        run_all(gens)
        c.synchronize()
        el = time.perf_counter() - t0
        b = prob.algorithmic_bytes_per_candidate(lam)
        rate = Q * lam * gens / el
        print("Q=%3d  %.3e candidate-trajectories/s  %.2f us per generation (all queries)  %.2f us per query  %.1f GB/s "
              "algorithmic (%.3f of 8 TB/s)" % (Q, rate, el / gens * 1e6, el / gens * 1e6 / Q, rate * b / 1e9,
                                               rate * b / 8e12), flush=True)
        if Q == 1:   # the bits the run ended in (to compare builds and switches)
            st = c.opt_get()
            print("       final state: sigma %r  best_f %r  sum(xmean) %r  itr %d" % (st["sigma"], st["best_f"],
                                                                                 float(st["xmean"].sum()), st["itr"]), flush=True)
        c.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "batch":
        return batch_main(sys.argv[2:])
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    ks = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8, 16]
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(cfg)
    lam = prob.lam if cfg <= 3 else prob.lam // 8
    episode = 100
    for K in ks:
        ctxs = []
        for q in range(K):
            c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42 + q,
                               start=prob.start, goal=prob.goal)
            c.set_grid(prob.cells)
            c.set_bounds(prob.lo, prob.hi)
            c.opt_init(prob.x0, prob.sigma0)
            c.run(12)
            ctxs.append(c)
        for c in ctxs:
            c.synchronize()

        def run_all(n):
            done = 0
            while done < n:
                step = min(episode, n - done)
                for c in ctxs:
                    c.opt_reset()
                    c.run(step)
                done += step

        run_all(200)
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        run_all(gens)
        for c in ctxs:
            c.synchronize()
        el = time.perf_counter() - t0
        b = prob.algorithmic_bytes_per_candidate(lam)
        rate = K * lam * gens / el
        print("K=%2d  %.3e candidate-trajectories/s  %.2f us per generation per query  %.1f GB/s algorithmic (%.3f of 8 TB/s)"
              % (K, rate, el / gens * 1e6, rate * b / 1e9, rate * b / 8e12), flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
