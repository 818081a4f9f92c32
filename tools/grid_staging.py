#!/usr/bin/env python3
"""Occupancy grid in LDS against lookups through L2: the measurements behind the choice.

    LMCMA_HIP_GRID_LDS=0|1 python tools/grid_staging.py run  <cfg2|vox64|vox64big> [generations]
    python tools/grid_staging.py spans

run: per-generation time of the whole loop with the bit-packed grid staged in each sampling
workgroup's LDS (1) or read from L2 (0); under rocprofv3 the kernel table shows the sampling kernel
of either variant.  cfg2 = the 512^2 map (32 KiB of bits); vox64 = a 64^3 voxel map (32 KiB, the
largest cube whose bits fit beside the rows and direction vectors) with cfg3's path and population
(W = 48, lambda = 8192, m = 32); vox64big = the same with lambda = 65536 (16 candidate tiles per
workgroup instead of 2).  The 128^3 map of cfg3 (256 KiB of bits) does not fit the 160 KiB of LDS,
so it could only be staged a slab at a time:
spans: for cfg3, how many of the four 32-layer z slabs (64 KiB of bits each) the waypoints of one
candidate, and of one workgroup's 8 candidates, touch after 0, 20 and 60 generations.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lmcma_path_planner_amd import grids, hip  # noqa: E402


def context(prob, lam):
    c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start,
                       goal=prob.goal)
    c.set_grid(prob.cells)
    c.set_bounds(prob.lo, prob.hi)
    c.opt_init(prob.x0, prob.sigma0)
    return c


def run(case, gens):
    if case == "cfg2":
        prob = grids.benchmark_problem(2)
        lam = prob.lam
    else:
        prob = grids.Problem("vox64-w48", 3, 64, 48, 8192, 32, seed=0xC0FFEE + 3)
        lam = 65536 if case == "vox64big" else 8192
    c = context(prob, lam)

    def steps(n):
        done = 0
        while done < n:
            k = min(100, n - done)
            c.opt_reset()
            c.run(k)
            done += k

    steps(112)
    c.synchronize()
    t0 = time.perf_counter()
    steps(gens)
    c.synchronize()
    el = time.perf_counter() - t0
    print("%s  LMCMA_HIP_GRID_LDS=%s  lambda %d  %.2f us per generation" % (
        case, os.environ.get("LMCMA_HIP_GRID_LDS", "unset"), lam, el / gens * 1e6))
    c.close()


def spans():
    prob = grids.benchmark_problem(3)
    c = context(prob, prob.lam)
    W, slab = prob.W, 32
    done = 0
    for at in (0, 20, 60):
        if at > done:
            c.run(at - done)
            done = at
        c.phase_local()   # this generation's candidates (the state is not advanced)
        rows = c.rows()
        z = np.rint(rows[:, 2 * W: 3 * W]).astype(int)
        z = np.concatenate([np.full((len(z), 1), int(prob.start[2])), z, np.full((len(z), 1), int(prob.goal[2]))], axis=1)
        touched = np.zeros((len(z), prob.size // slab), dtype=bool)
        for s in range(prob.size // slab):
            touched[:, s] = np.any((z // slab) == s, axis=1)
        per_cand = touched.sum(axis=1)
        per_wg = touched.reshape(-1, 8, touched.shape[1]).any(axis=1).sum(axis=1)
        print("generation %3d: z slabs of %d layers touched per candidate: min %d mean %.2f max %d; per workgroup "
              "of 8 candidates: min %d mean %.2f (of %d slabs)" % (at, slab, per_cand.min(), per_cand.mean(),
                                                                  per_cand.max(), per_wg.min(), per_wg.mean(),
                                                                  touched.shape[1]))
    c.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "spans":
        spans()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "cfg2", int(sys.argv[3]) if len(sys.argv) > 3 else 600)
