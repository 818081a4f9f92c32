"""Converged path cost, device loop (fp32 rows and costs) against the CPU oracle (fp64), on
obstacle maps.

On an obstacle-free map the cost has one optimum and the two runs agree to 1e-4
(test_gpu_parity.py::test_end_to_end_converged_path_cost_matches_cpu_oracle).  On an obstacle
map the grid term is a nearest-cell lookup (the reference's nearbyint, ...path_planner.cpp:
595-602): piecewise constant, so the search is chaotic at the scale of a cell, and two runs of
the SAME fp64 oracle whose initial mean differs by one float ulp end as far apart as the device
and the oracle do.  What can be asserted, and is, over many seeds:

  * control: the paired difference |J_device - J_oracle| / J_oracle is no larger (median, and on
    clearance-cost maps the upper quartile, factor 3) than the paired difference between the oracle and the oracle started one
    float ulp away: the device's rounding moves the result no more than any rounding does
    (measured: medians 9.5e-3 against 5.3e-3, 6.9e-2 against 1.2e-1, 7.1e-3 against 6.2e-3);
  * distribution: on clearance-cost maps median and quartiles of the final cost agree within 5 % /
    10 %; on occupancy maps the share of collision-free final paths agrees within 25 points and the
    median cost of the collision-free ones within 15 %;
  * the stated 1e-4 itself is asserted where it can hold and can fail: on an obstacle map whose
    optimum lies in free space (test_converged_cost_agrees_to_1e_4_where_the_optimum_is_in_free_space);
    on the maps above no two runs end on identical cells, so there is nothing to assert it on.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _final(oracle, hip, prob, cells, lam, m, gens, seed):
    """(oracle cost, device cost, cost of the oracle started one float ulp away), with the
    grid terms, and the two mean paths."""
    co = oracle.CostOracle(prob.dims, prob.W, cells, prob.start, prob.goal)

    def cpu(x0):
        o = oracle.OracleLMCMA(prob.N, lam, m, initial=x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=seed,
                               z_source=oracle.Z_PHILOX)
        o.run_trajectory(co, gens)
        return o.xmean

    x_cpu = cpu(prob.x0)
    x_ulp = cpu(prob.x0 + np.spacing(np.float32(prob.size / 2)).astype(np.float64))
    ctx = hip.HipContext(prob.N, lam, m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=seed)
    ctx.set_grid(cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)
    ctx.run(gens)
    x_gpu = ctx.opt_get()["xmean"]
    ctx.close()
    return co, co.cost(x_cpu), co.cost(x_gpu), co.cost(x_ulp), x_cpu, x_gpu


def _paired(a, b):
    return np.abs(a - b) / np.minimum(a, b)


@pytest.mark.parametrize("name,cfg,lam,m,gens,esdf,seeds", [("cfg1 clearance", 1, 16, 4, 1500, True, 32),
                                                            ("cfg1 occupancy", 1, 16, 4, 1500, False, 32),
                                                            ("cfg2 cut clearance", 2, 256, 16, 800, True, 12)])
def test_converged_cost_on_obstacle_maps_device_vs_oracle(oracle, name, cfg, lam, m, gens, esdf, seeds):
    from lmcma_path_planner_amd import grids, hip
    oracle.set_threads(8)
    try:
        prob = grids.benchmark_problem(cfg)
        cells = grids.state_cost_from_distance(grids.edt(prob.occ), 0.5) if esdf else prob.occ
        res = [_final(oracle, hip, prob, cells, lam, m, gens, seed) for seed in range(1, seeds + 1)]
    finally:
        oracle.set_threads(1)
    co = res[0][0]
    straight, _ = co.cost(prob.x0)
    cpu = np.array([r[1][0] for r in res])
    gpu = np.array([r[2][0] for r in res])
    ulp = np.array([r[3][0] for r in res])
    cpu_grid = np.array([r[1][1][1] for r in res])
    gpu_grid = np.array([r[2][1][1] for r in res])
    d_dev, d_ctl = _paired(gpu, cpu), _paired(ulp, cpu)
    q = lambda v: np.percentile(v, [25, 50, 75])
    print("%s: final cost q25/50/75 oracle %s device %s; paired difference median/q75 device-oracle %.2e/%.2e, "
          "oracle-oracle(+1 ulp) %.2e/%.2e" % (name, q(cpu).round(1), q(gpu).round(1), np.median(d_dev),
                                               np.percentile(d_dev, 75), np.median(d_ctl), np.percentile(d_ctl, 75)))
    # every run improves on the straight line, the typical one by a wide margin
    assert gpu.max() < straight and cpu.max() < straight
    assert np.median(gpu) < 0.5 * straight and np.median(cpu) < 0.5 * straight
    # control: the device's rounding moves the result no more than a one-ulp change of the start does
    # (factor 3 on the median and on the upper quartile of 12-32 paired differences)
    assert np.median(d_dev) <= 3.0 * np.median(d_ctl) + 1e-4
    if esdf:
        # (on an occupancy map the final costs are bimodal, collision or none, a factor of ten apart,
        # and the upper quartile of the paired differences only says how many pairs straddle the modes;
        # the shares of the modes are compared below instead)
        assert np.percentile(d_dev, 75) <= 3.0 * np.percentile(d_ctl, 75) + 1e-4
    if esdf:
        qc, qg = q(cpu), q(gpu)
        assert abs(qg[1] - qc[1]) <= 0.05 * qc[1]
        assert abs(qg[0] - qc[0]) <= 0.10 * qc[0] and abs(qg[2] - qc[2]) <= 0.10 * qc[2]
    else:
        free_c, free_g = cpu_grid == 0.0, gpu_grid == 0.0
        assert abs(free_c.mean() - free_g.mean()) <= 0.25
        assert free_c.sum() >= 8 and free_g.sum() >= 8
        mc, mg = np.median(cpu[free_c]), np.median(gpu[free_g])
        assert abs(mg - mc) <= 0.15 * mc
    same = sum(np.array_equal(np.rint(r[4].astype(np.float32)), np.rint(r[5].astype(np.float32))) for r in res)
    print("%s: %d of %d seeds end on identical cells" % (name, same, seeds))


def _free_diagonal_map(size, reach):
    """Two square obstacles in the corners the straight start-goal line does not visit."""
    occ = np.zeros((size, size), dtype=np.uint8)
    occ[:reach, :reach] = 1
    occ[size - reach:, size - reach:] = 1
    return occ


def test_converged_cost_agrees_to_1e_4_where_the_optimum_is_in_free_space(oracle):
    """cfg1's geometry (100 x 100 cells, 32 waypoints, start (99, 0), goal (0, 99), bounds, sigma0 = 5)
    with obstacles ON the map and IN the way of the search, but an optimum interior to free space:
    two 44-cell squares fill the corners beside the diagonal, and the initial path bows 14 cells into
    one of them (its cost is 25 674 against 140 for the optimum; two thirds of the first ten
    generations' candidates touch an obstacle).  The search has to leave the obstacle; once it has,
    length and smoothness have one optimum, and the stated criterion applies: the converged cost of
    the device loop (fp32 rows and costs) and of the fp64 oracle agree to 1e-4 relative.  Asserted:
    at least 75 % of 16 seeds end collision-free on both sides AND agree to 1e-4, and the typical
    difference is far below the bound (measured: 16 of 16 collision-free, 15 agree, median 2e-5, worst
    1.4e-4).  What is left is the float floor, not a different optimum: the device's costs are floats
    near 140 (one ulp = 1.5e-5), so candidates closer together than that rank by index and its step
    size settles near 3e-4 where the oracle's keeps shrinking; the device's converged cost stays within
    one or two 1e-4 above the fp64 optimum.  (With a 1 / clearance map that is zero from 4 cells on, the
    ORACLE ends in a second optimum, 142 to 145, in some seeds; that variant is not asserted.)"""
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(1)
    occ = _free_diagonal_map(100, 44)
    kind, cells = "occupancy", occ
    t = np.arange(1, prob.W + 1) / (prob.W + 1)
    x0 = prob.x0 - np.concatenate([14 * np.sin(np.pi * t), 14 * np.sin(np.pi * t)])
    lam, m, gens, seeds = 32, 8, 8000, 16
    co = oracle.CostOracle(2, prob.W, cells, prob.start, prob.goal)
    start_cost, start_terms = co.cost(x0)
    assert start_terms[1] > 20.0 and start_cost > 2.0e4          # the initial path crosses an obstacle
    oracle.set_threads(8)
    try:
        rel, free = [], []
        for seed in range(1, seeds + 1):
            o = oracle.OracleLMCMA(prob.N, lam, m, initial=x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=seed,
                                   z_source=oracle.Z_PHILOX)
            o.run_trajectory(co, gens)
            ctx = hip.HipContext(prob.N, lam, m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=seed)
            ctx.set_grid(cells)
            ctx.set_bounds(prob.lo, prob.hi)
            ctx.opt_init(x0, prob.sigma0)
            ctx.run(gens)
            x_gpu = ctx.opt_get()["xmean"]
            ctx.close()
            (jc, tc), (jg, tg) = co.cost(o.xmean), co.cost(x_gpu)
            rel.append(abs(jc - jg) / jc)
            free.append(tc[1] == 0.0 and tg[1] == 0.0)
    finally:
        oracle.set_threads(1)
    rel, free = np.array(rel), np.array(free)
    agree = free & (rel <= 1e-4)
    print("%s map, optimum in free space: %d of %d seeds collision-free on both sides, %d agree to 1e-4; relative "
          "difference median %.2e, worst among the agreeing %.2e, worst %.2e"
          % (kind, free.sum(), seeds, agree.sum(), np.median(rel), rel[agree].max() if agree.any() else np.nan, rel.max()))
    assert agree.sum() >= 0.75 * seeds
    assert np.median(rel) <= 5e-5


def test_cfg2_at_its_stated_shape_share_of_collision_free_paths(oracle):
    """cfg2 as the benchmark runs it (512 x 512 map with two bars and random boxes, 64 waypoints,
    lambda = 4096, m = 16, one episode of 100 generations from the straight line): the share of
    seeds whose final mean path is collision-free, device loop and fp64 oracle.  The straight line
    crosses obstacles (cost about 1e5), and 100 generations do not always find a way round: the
    shares are reported, and must be alike (within 3 of 8 seeds) and not empty."""
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(2)
    co = oracle.CostOracle(2, prob.W, prob.cells, prob.start, prob.goal)
    straight, straight_terms = co.cost(prob.x0)
    assert straight_terms[1] > 0.0
    oracle.set_threads(8)
    try:
        free_c = free_g = 0
        costs = []
        for seed in range(1, 9):
            o = oracle.OracleLMCMA(prob.N, prob.lam, prob.m, initial=prob.x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0,
                                   seed=seed, z_source=oracle.Z_PHILOX)
            o.run_trajectory(co, 100)
            ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal,
                                 seed=seed)
            ctx.set_grid(prob.cells)
            ctx.set_bounds(prob.lo, prob.hi)
            ctx.opt_init(prob.x0, prob.sigma0)
            ctx.run(100)
            x_gpu = ctx.opt_get()["xmean"]
            ctx.close()
            (jc, tc), (jg, tg) = co.cost(o.xmean), co.cost(x_gpu)
            free_c += tc[1] == 0.0
            free_g += tg[1] == 0.0
            costs.append((round(jc), round(jg)))
            assert jc < straight and jg < straight
    finally:
        oracle.set_threads(1)
    print("cfg2 at its stated shape, 100 generations: collision-free final mean paths oracle %d / 8, device %d / 8; "
          "straight line %.0f; final costs (oracle, device) %s" % (free_c, free_g, straight, costs))
    assert free_c >= 2 and free_g >= 2 and abs(int(free_c) - int(free_g)) <= 3


def test_step_ranks_float_costs_like_the_fp64_oracle_except_within_rounding(oracle):
    """LMCMA::step() in its host-update mode ranks the device's float costs (the reference ranks
    doubles, lmcma.cpp:315, 400).  On a population with many near-ties, every pair of candidates the
    device orders differently from the fp64 oracle (same rows) has fp64 costs within the stated cost
    tolerance 1e-5 max(1, |J|) of each other; exact float ties keep candidate order; and the mean
    recombined from the device's ranking differs from the oracle's by no more than those pairs allow."""
    from lmcma_path_planner_amd import LMCMA, grids
    prob = grids.benchmark_problem(1)
    lam, m = 512, 8
    co = oracle.CostOracle(2, prob.W, prob.occ, prob.start, prob.goal)
    # a narrow population around the straight line: sigma 1e-3 cells, so costs differ in the 6th-8th digit
    opt = LMCMA(prob.x0, lam, prob.lo, prob.hi, 1e-3, None, 42)
    opt.setOptions(m=m, philox=True)
    opt.setTrajectoryProblem(2, prob.W, prob.occ, prob.start, prob.goal)
    opt.init(prob.N)
    flips_total = 0
    for g in range(6):
        rows = opt.population()
        opt.step(1)
        got = opt.lastCosts()                                    # float costs, as the device ranked them
        want, _ = co.cost_rows_f32(rows.astype(np.float32))      # fp64 costs of the same rows
        tol = 1e-5 * np.maximum(1.0, np.abs(want))
        assert np.all(np.abs(got - want) <= tol)
        dev_order = np.argsort(got.astype(np.float32), kind="stable")
        ref_order = np.argsort(want, kind="stable")
        dev_rank = np.empty(lam, dtype=np.int64)
        dev_rank[dev_order] = np.arange(lam)
        # walk the oracle's order: whenever the device puts a later candidate first, the two are near-tied
        for a, b in zip(ref_order[:-1], ref_order[1:]):
            if dev_rank[a] > dev_rank[b]:
                flips_total += 1
                assert abs(want[a] - want[b]) <= tol[a] + tol[b], (g, a, b, want[a], want[b])
        # exact float ties keep candidate order
        g32 = got.astype(np.float32)
        tied = np.flatnonzero(g32[dev_order][1:] == g32[dev_order][:-1])
        assert np.all(dev_order[tied] < dev_order[tied + 1])
    print("near-tie ranking: %d adjacent pairs ordered differently from fp64 over 6 generations, all within tolerance"
          % flips_total)
    opt.close()
