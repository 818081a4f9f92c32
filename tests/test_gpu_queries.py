"""Several independent optimisers ("queries") in one context: every launch advances all of them.

Query q of a context with seed s must make exactly the run a one-query context with seed s + q
makes (same kernels, same arithmetic, its own copies of every buffer): bit-identical state.  And a
query followed phase by phase must agree with the CPU oracle like a one-query context does.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(hip, prob, lam, m, seed, queries=1):
    ctx = hip.HipContext(prob.N, lam, m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=seed,
                         queries=queries)
    ctx.set_grid(prob.occ)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)
    return ctx


@pytest.mark.parametrize("cfg,lam,m,Q,gens", [(2, 512, 16, 5, 27),      # two-part update riding in the ranking launch
                                              (2, 4096, 16, 3, 19),     # the headline shape
                                              (3, 8192, 32, 2, 37),     # one-kernel update, two-level row sums
                                              (1, 16, 4, 7, 40)])       # tiny population
def test_queries_in_one_context_equal_separate_contexts(cfg, lam, m, Q, gens):
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(cfg)
    seed = 1000
    rng = np.random.default_rng(cfg)
    starts = [np.clip(prob.x0 + (rng.normal(scale=prob.sigma0, size=prob.N) if q % 2 else 0.0), prob.lo, prob.hi)
              for q in range(Q)]
    sigmas = [prob.sigma0 * (1.0 + 0.25 * q) for q in range(Q)]
    batch = _make(hip, prob, lam, m, seed, queries=Q)
    for q in range(Q):
        batch.opt_set_query(q, starts[q], sigmas[q])
    batch.run(gens)
    for q in range(Q):
        one = _make(hip, prob, lam, m, seed + q)
        one.opt_set_query(0, starts[q], sigmas[q])
        one.run(gens)
        a = one.opt_get()
        batch.query_select(q)
        b = batch.opt_get()
        assert a["itr"] == b["itr"] == gens and a["counteval"] == b["counteval"] == gens * lam
        np.testing.assert_array_equal(a["xmean"], b["xmean"])
        assert a["sigma"] == b["sigma"] and a["best_f"] == b["best_f"] and a["s"] == b["s"]
        va, vb = one.opt_vectors(), batch.opt_vectors()
        for key in ("pc", "v_arr", "pc_arr", "nj", "lj", "iterator"):
            np.testing.assert_array_equal(va[key], vb[key])
        np.testing.assert_array_equal(one.rows(), batch.rows())
        np.testing.assert_array_equal(one.costs(), batch.costs())
        one.close()
    # queries differ from each other (own seeds, own starts)
    batch.query_select(0)
    x0 = batch.opt_get()["xmean"]
    batch.query_select(1)
    assert not np.array_equal(x0, batch.opt_get()["xmean"])
    # a new episode: every query back to its own start
    batch.opt_reset()
    batch.run(3)
    for q in range(Q):
        batch.query_select(q)
        st = batch.opt_get()
        assert st["itr"] == 3 and not st["done"]
    batch.close()


def test_a_query_of_a_batch_follows_the_oracle_phase_by_phase(oracle):
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(1)
    lam, m, Q, seed = 64, 8, 3, 7
    ctx = _make(hip, prob, lam, m, seed, queries=Q)
    co = oracle.CostOracle(2, prob.W, prob.occ, prob.start, prob.goal)
    os_ = [oracle.OracleLMCMA(prob.N, lam, m, initial=prob.x0, lo=prob.lo, hi=prob.hi, sigma=prob.sigma0, seed=seed + q,
                              z_source=oracle.Z_GIVEN) for q in range(Q)]
    for g in range(m + 6):
        ctx.phase_local()
        got = []
        for q in range(Q):
            ctx.query_select(q)
            rows, costs = ctx.rows(), ctx.costs()
            if g == 0:   # query q draws from the Philox key seed + q
                z = np.stack([oracle.philox_normals(seed + q, 0, i, prob.N) for i in range(lam)])
                want = np.clip(prob.x0 + prob.sigma0 * z, prob.lo, prob.hi)
                assert np.max(np.abs(rows - want)) <= 2e-5 * (prob.size + 6 * prob.sigma0)
            want, _ = co.cost_rows_f32(rows.astype(np.float32))
            assert np.max(np.abs(costs - want) / np.maximum(1.0, np.abs(want))) <= 1e-5
            got.append((rows, costs))
        ctx.phase_rank()
        ctx.phase_update(0)
        for q in range(Q):
            ctx.query_select(q)
            o = os_[q]
            o.set_arx(got[q][0])
            o.update_only(got[q][1])
            st, vec = ctx.opt_get(), ctx.opt_vectors()
            assert st["itr"] == o.itr == g + 1 and st["iterator_sz"] == o.iterator_sz
            np.testing.assert_array_equal(vec["iterator"][: o.iterator_sz], o.iterator[: o.iterator_sz])
            np.testing.assert_allclose(st["xmean"], o.xmean, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(st["sigma"], o.sigma, rtol=1e-12)
            live = o.iterator[: o.iterator_sz]
            np.testing.assert_allclose(vec["v_arr"][live], o.v_arr[live], rtol=1e-9, atol=1e-11)
            assert st["best_f"] == o.BestF and st["counteval"] == o.counteval
    ctx.close()


def test_stop_rule_freezes_each_query_on_its_own(oracle):
    from lmcma_path_planner_amd import hip
    W, N, lam, m = 8, 16, 32, 4
    lo, hi = np.full(N, -1.0), np.full(N, 1.0)
    ctx = hip.HipContext(N, lam, m, dims=2, waypoints=W, start=[0.0, 0.0], goal=[0.0, 0.0], seed=5, queries=3)
    ctx.set_grid(np.zeros((8, 8), dtype=np.uint8))
    ctx.set_bounds(lo, hi)
    ctx.opt_init(np.zeros(N), 0.1)
    ctx.opt_set_query(1, np.zeros(N), 2e-19)       # this one crosses sigma < 1e-20 within a few dozen generations
    ctx.run(150)
    states = []
    for q in range(3):
        ctx.query_select(q)
        states.append(ctx.opt_get())
    assert states[1]["done"] and states[1]["itr"] < 150 and states[1]["sigma"] < 1e-20
    assert not states[0]["done"] and not states[2]["done"] and states[0]["itr"] == states[2]["itr"] == 150
    # the same query alone stops in the same generation
    one = hip.HipContext(N, lam, m, dims=2, waypoints=W, start=[0.0, 0.0], goal=[0.0, 0.0], seed=6)
    one.set_grid(np.zeros((8, 8), dtype=np.uint8))
    one.set_bounds(lo, hi)
    one.opt_init(np.zeros(N), 2e-19)
    one.run(150)
    st = one.opt_get()
    assert st["done"] and st["itr"] == states[1]["itr"] and st["sigma"] == states[1]["sigma"]
    one.close()
    ctx.close()


def test_queries_argument_checks():
    from lmcma_path_planner_amd import grids, hip
    from lmcma_path_planner_amd._lib import LmcmaError
    prob = grids.benchmark_problem(1)
    with pytest.raises(LmcmaError, match="several queries"):
        hip.HipContext(prob.N, 32768, 4, dims=2, waypoints=prob.W, queries=2)          # beyond the counting rank
    with pytest.raises(LmcmaError, match="several queries"):
        hip.HipContext(prob.N, 64, 4, dims=2, waypoints=prob.W, lambda_local=32, queries=2)
    with pytest.raises(LmcmaError, match="queries must be"):
        hip.HipContext(prob.N, 64, 4, dims=2, waypoints=prob.W, queries=257)
    ctx = _make(hip, prob, 64, 4, 1, queries=2)
    with pytest.raises(LmcmaError, match="one query per context"):
        ctx.select()
    with pytest.raises(LmcmaError, match="one query per context"):
        ctx.evaluate()
    with pytest.raises(LmcmaError, match="no such query"):
        ctx.query_select(2)
    with pytest.raises(LmcmaError, match="not a shard"):
        ctx.phase_update(2)
    ctx.close()


@pytest.mark.parametrize("lam,mode,gens", [(4096, "random", 20), (2048, "random", 6), (1024, "ties", 4), (4096, "ties", 3),
                                           (1024, "3d", 12),
                                           (1000, "random", 5)])    # counting rank, a last tile of 40 candidates
def test_many_queries_are_ranked_one_workgroup_each(oracle, lam, mode, gens):
    """From 12 queries on, every population of a context is ranked by one workgroup (a sample sort)
    and its elite rows are added by the rank-ordered recombination launch.  Checked per generation
    for three of the queries: the permutation and the sorted costs are those of a stable sort of the
    query's costs ("ties": degenerate bounds make all costs equal, so the order is the candidate
    order), and the state follows the CPU oracle handed the same rows and costs.  The replayed
    loop of a second context makes the same run bit for bit, and a one-query context (ranked by
    counting, another order of the same fp64 sums) agrees to rounding."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    from lmcma_path_planner_amd import grids, hip
    Q, m, seed = 16, 16, 31
    if mode == "3d":
        # rows shorter than the 128 floats of the matrix-core tiles (30 of 32 chunks), three
        # dimensions, a float clearance-cost grid, fewer vectors than a tile has rows
        prob = grids.Problem("t3d", dims=3, size=40, waypoints=40, lam=lam, m=9, esdf=True)
        m, mode = 9, "random"
    else:
        prob = grids.benchmark_problem(2)
    dev = torch.device("cuda", 0)
    lo, hi = (prob.x0, prob.x0) if mode == "ties" else (prob.lo, prob.hi)

    def make(queries, s):
        c = hip.HipContext(prob.N, lam, m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=s,
                           queries=queries)
        c.set_grid(prob.cells)
        c.set_bounds(lo, hi)
        c.opt_init(prob.x0, prob.sigma0)
        return c

    ctx = make(Q, seed)
    watch = (0, 7, 15)
    os_ = {q: oracle.OracleLMCMA(prob.N, lam, m, initial=prob.x0, lo=lo, hi=hi, sigma=prob.sigma0, seed=seed + q,
                                 z_source=oracle.Z_GIVEN) for q in watch}
    co = oracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
    M = np.sqrt(1.0 - 1.0 / (10.0 * np.log(prob.N + 1.0)))      # lmcma.cpp:268, 431-447
    some = np.r_[0:48, lam - 16:lam]
    for g in range(gens):
        before = {}
        if mode == "random" and g in (0, 3, gens - 1):
            for q in watch:
                ctx.query_select(q)
                before[q] = (ctx.opt_get(), ctx.opt_vectors())
        ctx.phase_local()
        got = {}
        for q in watch:
            ctx.query_select(q)
            got[q] = (ctx.rows(), ctx.costs())
            if q in before:
                # the rows (sampled on the matrix cores from 16 queries on) against computeAz in
                # closed form, fp64, from the state before the launch; the costs against the oracle
                st, vec = before[q]
                kk = st["iterator_sz"]
                slots = vec["iterator"][:kk]
                z = np.stack([oracle.philox_normals(seed + q, g, int(i), prob.N) for i in some])
                az = M ** kk * z
                for t, sl in enumerate(slots):
                    az += (M ** (kk - 1 - t) * vec["nj"][sl]) * np.outer(z @ vec["v_arr"][sl], vec["pc_arr"][sl])
                want = np.clip(st["xmean"] + st["sigma"] * az, lo, hi)
                assert np.max(np.abs(got[q][0][some] - want)) <= 2e-5 * (prob.size + 6 * prob.sigma0)
                wc, _ = co.cost_rows_f32(got[q][0][some].astype(np.float32))
                assert np.max(np.abs(got[q][1][some] - wc) / np.maximum(1.0, np.abs(wc))) <= 1e-5
        ctx.phase_rank()
        ctx.synchronize()
        for q in watch:
            ctx.query_select(q)
            raw = got[q][1].astype(np.float32)
            order = np.argsort(raw.astype(np.float64), kind="stable")
            perm = shard_mod.alias_tensor(ctx, hip.BUF_PERM, torch.int32, dev).cpu().numpy()
            sc = shard_mod.alias_tensor(ctx, hip.BUF_SORTED_COST, torch.float32, dev).cpu().numpy()
            np.testing.assert_array_equal(perm, order)
            np.testing.assert_array_equal(sc, raw[order])
            if mode == "ties":
                assert np.all(raw == raw[0])
        ctx.phase_update(0)
        for q in watch:
            ctx.query_select(q)
            o = os_[q]
            o.set_arx(got[q][0])
            o.update_only(got[q][1])
            st, vec = ctx.opt_get(), ctx.opt_vectors()
            assert st["itr"] == o.itr == g + 1 and st["iterator_sz"] == o.iterator_sz
            np.testing.assert_allclose(st["xmean"], o.xmean, rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(st["sigma"], o.sigma, rtol=1e-11)
            assert st["best_f"] == o.BestF and st["counteval"] == o.counteval
            live = o.iterator[: o.iterator_sz]
            np.testing.assert_allclose(vec["v_arr"][live], o.v_arr[live], rtol=1e-8, atol=1e-10)
    # the replayed loop: same launches, same bits
    again = make(Q, seed)
    again.run(gens)
    for q in watch:
        ctx.query_select(q)
        again.query_select(q)
        a, b = ctx.opt_get(), again.opt_get()
        np.testing.assert_array_equal(a["xmean"], b["xmean"])
        assert a["sigma"] == b["sigma"] and a["itr"] == b["itr"] == gens
    # a context of its own per query samples on the vector units and ranks by counting: the same
    # normals, sums in another order, so its first generation's costs agree to rounding
    one = make(1, seed + 7)
    one.phase_local()
    again.opt_reset()
    again.phase_local()
    again.query_select(7)
    np.testing.assert_allclose(again.rows(), one.rows(), rtol=0, atol=2e-5 * (prob.size + 6 * prob.sigma0))
    np.testing.assert_allclose(again.costs(), one.costs(), rtol=1e-4)
    for c in (ctx, again, one):
        c.close()
