#!/usr/bin/env python3
"""How far the device's direction vectors and the fp64 oracle's drift apart at a benchmark
shape, and how far each is from the same recurrence in extended precision.

    python tools/update_conditioning.py [cfg] [generations]

Per generation: the largest |v_device - v_oracle|, |v_device - v_ext| and |v_oracle - v_ext|
over the live vectors, each relative to the largest entry of the vectors.  v_ext rebuilds
every live vector from the oracle's stored paths with numpy longdouble (x87 80-bit) in the
reference's order of operations (lmcma.cpp:375-390, 449-463).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rebuild_ext(pc_arr, iterator, k, c1):
    """v_i = T_{i-1} ... T_0 pc_i for every live position, in longdouble."""
    ld = np.longdouble
    c1 = ld(c1)
    K = 1 / np.sqrt(1 - c1)
    sq = np.sqrt(1 - c1)
    ratio = c1 / (1 - c1)
    V, L = [], []
    for i in range(k):
        a = pc_arr[iterator[i]].astype(ld)
        for j in range(i):
            a = K * a - L[j] * np.dot(V[j], a) * V[j]
        nv = np.dot(a, a)
        V.append(a)
        L.append((1 / (sq * nv)) * (1 - 1 / np.sqrt(1 + ratio * nv)))
    return np.stack(V)


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 44
    from lmcma_path_planner_amd import grids, hip
    from oracle import pyoracle as oracle
    prob = grids.benchmark_problem(cfg)
    lam = prob.lam if cfg <= 3 else prob.lam // 8
    N, m = prob.N, prob.m
    ctx = hip.HipContext(N, lam, m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)
    o = oracle.OracleLMCMA(N, lam, m, initial=prob.x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=42,
                           z_source=oracle.Z_GIVEN)
    print("gen  k   dev-orc    dev-ext    orc-ext   (relative to the largest entry)")
    for g in range(gens):
        ctx.phase_local()
        rows, costs = ctx.rows(), ctx.costs()
        ctx.phase_rank()
        ctx.phase_update(0)
        st = ctx.opt_get()
        o.set_arx(rows)
        o.update_with_mean(costs, st["xmean"])
        vec = ctx.opt_vectors()
        k = o.iterator_sz
        it = o.iterator[:k]
        vo, vd = o.v_arr[it], vec["v_arr"][it]
        # the oracle only rebuilds from imin on; the extended-precision chain rebuilds all, from the same paths
        ve = rebuild_ext(o.pc_arr, it, k, o.c1)
        big = float(np.abs(vo).max())
        print("%3d %3d  %.2e  %.2e  %.2e" % (g, k, np.abs(vd - vo).max() / big,
                                             float(np.abs(vd - ve).max()) / big, float(np.abs(vo - ve).max()) / big))
    ctx.close()


if __name__ == "__main__":
    main()
