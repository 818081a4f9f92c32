"""Parity of the HIP path with the CPU oracle, through the C ABI (ctypes).

Tolerances (fp32 device arithmetic against the fp64 oracle):
  sampled coordinate      |dx| <= 2e-6 * (|xmean| + sigma * |A z|_bound)   (host-supplied z)
                          |dx| <= 2e-5 * that bound                        (Philox: fp32 log/sincos)
  per-candidate cost      |dJ| / max(1, |J|) <= 1e-5 on identical fp32 rows
  ranking / permutation   bit-exact
  fp64 update state       relative 1e-11 (summation order differs, nothing else)
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lmcma_path_planner_amd import hip as hipmod
    return hipmod


def _evolved_oracle(oracle, N, lam, m, gens, seed=11, sigma=2.0, lo=None, hi=None):
    """An oracle optimiser that has run `gens` generations on a shifted ellipsoid,
    so that the direction store holds min(gens, m) live vectors."""
    scale = 10.0 ** (2.0 * np.arange(N) / max(N - 1, 1))
    o = oracle.OracleLMCMA(N, lam, m, initial=np.full(N, 3.0), lo=lo, hi=hi, sigma=sigma, seed=seed)
    for _ in range(gens):
        X = o.arx
        o.tell_all(np.sum(scale * (X - 1.0) ** 2, axis=1))
    return o


def _state_of(o):
    k = o.iterator_sz
    it = o.iterator[:k]
    M = np.sqrt(1.0 - o.c1)
    return dict(xmean=o.xmean, sigma=o.sigma, V=o.v_arr[it], PC=o.pc_arr[it], nj=o.nj[it], M=M, k=k)


@pytest.mark.parametrize("N,lam,m,gens", [(128, 256, 16, 24), (64, 16, 4, 9), (144, 64, 32, 5), (6, 8, 8, 12),
                                          (288, 32, 32, 40)])
def test_sample_with_host_z_matches_oracle(hip, oracle, N, lam, m, gens):
    lo, hi = np.full(N, 0.25), np.full(N, 3.25)
    o = _evolved_oracle(oracle, N, lam, m, gens, lo=lo, hi=hi)
    st = _state_of(o)
    assert st["k"] == min(gens, m)
    rng = np.random.default_rng(5)
    Z = rng.normal(size=(lam, N))
    Zf = Z.astype(np.float32).astype(np.float64)   # what the device will see
    o.sample_given(Zf)
    want = o.arx
    ctx = hip.HipContext(N, lam, m, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    ctx.set_bounds(lo, hi)
    ctx.set_state(st["xmean"], st["sigma"], st["V"] if st["k"] else None, st["PC"] if st["k"] else None,
                  st["nj"] if st["k"] else None, st["M"])
    ctx.sample(generation=gens, z=Z)
    got = ctx.rows()
    bound = np.abs(st["xmean"]).max() + st["sigma"] * np.abs(want - st["xmean"]).max() / max(st["sigma"], 1e-30) + 1.0
    assert np.max(np.abs(got - want)) <= 2e-6 * bound
    # clamping is exact: anything the oracle put on a bound is on the same bound
    np.testing.assert_array_equal(got[want == hi[0]], hi[0])
    np.testing.assert_array_equal(got[want == lo[0]], lo[0])
    assert np.any(want == hi[0]) or np.any(want == lo[0])
    ctx.close()


def test_sample_with_philox_matches_oracle(hip, oracle):
    N, lam, m, gens, seed = 128, 512, 16, 20, 1234567890123
    o = _evolved_oracle(oracle, N, lam, m, gens)
    st = _state_of(o)
    first = 96   # a shard in the middle of the population
    nloc = 160
    Z = np.stack([oracle.philox_normals(seed, gens, first + i, N) for i in range(nloc)])
    Zfull = np.zeros((lam, N))
    Zfull[:nloc] = Z
    o.sample_given(Zfull)
    want = o.arx[:nloc]
    ctx = hip.HipContext(N, lam, m, lambda_local=nloc, first_candidate=first, z_source=hip.Z_PHILOX,
                         cost_source=hip.COST_EXTERNAL, seed=seed)
    ctx.set_state(st["xmean"], st["sigma"], st["V"], st["PC"], st["nj"], st["M"])
    ctx.sample(generation=gens)
    got = ctx.rows()
    scale = np.abs(st["xmean"]).max() + np.abs(want - st["xmean"]).max() + 1.0
    assert np.max(np.abs(got - want)) <= 2e-5 * scale
    # the generated normals themselves: k = 0, sigma = 1, xmean = 0 gives x = z
    ctx.set_state(np.zeros(N), 1.0, None, None, None, 1.0)
    ctx.sample(generation=7)
    z_dev = ctx.rows()
    z_ref = np.stack([oracle.philox_normals(seed, 7, first + i, N) for i in range(nloc)])
    assert np.max(np.abs(z_dev - z_ref)) <= 4e-6
    ctx.close()


@pytest.mark.parametrize("N,dims,W,factor", [(128, 2, 64, "smooth"), (144, 3, 48, "smooth"), (6, 2, 3, "smooth"),
                                               (33, 3, 11, "dense"), (288, 3, 96, "smooth"), (512, 2, 256, "band")])
def test_sample_with_structured_covariance_matches_oracle(hip, oracle, N, dims, W, factor):
    """z <- L z on the matrix cores (fp32) against applyCovL in fp64 on the same Philox normals.
    Tolerance: 1e-5 of the largest |L z| (sums of up to N fp32 products), then the usual
    2e-5 bound on the rows."""
    lam, m, gens, seed, first, nloc = 96, 8, 11, 987654321, 16, 72
    rng = np.random.default_rng(3)
    if factor == "smooth":
        L = oracle.cholesky(oracle.covariance(dims, W))           # column-major lower factor, as cholesky() leaves it
    elif factor == "dense":
        L = rng.normal(size=(N, N)) / np.sqrt(N)                   # not triangular: every block of every row counts
    else:
        Lm = np.tril(rng.normal(size=(N, N))) * (np.abs(np.subtract.outer(np.arange(N), np.arange(N))) < 40)
        L = np.ascontiguousarray(Lm.T)                             # L[c*N + r] = Lm(r, c)
    Lm = L.reshape(N, N).T
    Z0 = np.stack([oracle.philox_normals(seed, gens, first + i, N) for i in range(nloc)])
    Zc = Z0 @ Lm.T
    np.testing.assert_allclose(oracle.apply_cov_l(L, Z0[0]), Zc[0], rtol=1e-12, atol=1e-12)
    ctx = hip.HipContext(N, lam, m, lambda_local=nloc, first_candidate=first, z_source=hip.Z_PHILOX,
                         cost_source=hip.COST_EXTERNAL, seed=seed)
    ctx.set_cov_factor(L)
    # k = 0, sigma = 1, xmean = 0: the rows are L z themselves
    ctx.set_state(np.zeros(N), 1.0, None, None, None, 1.0)
    ctx.sample(generation=gens)
    assert np.max(np.abs(ctx.rows() - Zc)) <= 1e-5 * max(1.0, np.abs(Zc).max())
    # with an evolved direction store on top
    o = _evolved_oracle(oracle, N, lam, m, gens)
    st = _state_of(o)
    Zfull = np.zeros((lam, N))
    Zfull[:nloc] = Zc
    o.sample_given(Zfull)
    want = o.arx[:nloc]
    ctx.set_state(st["xmean"], st["sigma"], st["V"], st["PC"], st["nj"], st["M"])
    ctx.sample(generation=gens)
    scale = np.abs(st["xmean"]).max() + np.abs(want - st["xmean"]).max() + 1.0
    assert np.max(np.abs(ctx.rows() - want)) <= 2e-5 * scale
    # back to the identity
    ctx.set_cov_factor(None)
    ctx.set_state(np.zeros(N), 1.0, None, None, None, 1.0)
    ctx.sample(generation=gens)
    assert np.max(np.abs(ctx.rows() - Z0)) <= 4e-6
    ctx.close()


def test_device_generations_with_structured_covariance_follow_the_oracle(hip, oracle):
    """The whole device loop (Philox normals, L z, sample, score, rank, update) with a factor
    set, against the oracle driven with the device's own costs generation by generation."""
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(1)
    N, lam, m = prob.N, 64, 8
    cov = oracle.covariance(2, prob.W)
    L = oracle.cholesky(cov)
    sigma0 = prob.sigma0 * 8.0
    ctx = hip.HipContext(N, lam, m, dims=2, waypoints=prob.W, seed=42, start=prob.start, goal=prob.goal)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.set_cov_factor(L)
    ctx.opt_init(prob.x0, sigma0)
    o = oracle.OracleLMCMA(N, lam, m, initial=prob.x0, lo=prob.lo, hi=prob.hi, sigma=sigma0, cov=cov, seed=42,
                           z_source=oracle.Z_PHILOX)
    co = oracle.CostOracle(2, prob.W, prob.cells, prob.start, prob.goal)
    for g in range(12):
        ctx.phase_local()
        rows = ctx.rows()
        scale = np.abs(o.xmean).max() + np.abs(o.arx - o.xmean).max() + 1.0
        assert np.max(np.abs(rows - o.arx)) <= 4e-5 * scale, g
        want, _ = co.cost_rows_f32(rows.astype(np.float32))
        got = ctx.costs()
        assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-5
        ctx.phase_rank()
        ctx.phase_update(0)
        # the oracle continues from the device's rows and costs, so that rounding in the rows does not
        # send the two runs down different rankings
        o.set_arx(rows)
        o.tell_all(got)
        st = ctx.opt_get()
        np.testing.assert_allclose(st["xmean"], o.xmean, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(st["sigma"], o.sigma, rtol=1e-11)
    # and the same generations from one graph launch sequence
    ctx.opt_reset()
    ctx.run(12)
    st2 = ctx.opt_get()
    np.testing.assert_array_equal(st2["xmean"], st["xmean"])
    ctx.close()


def _random_rows(rng, prob, count, spread):
    base = prob.x0 + rng.normal(scale=spread, size=(count, prob.N))
    return np.clip(base, prob.lo, prob.hi).astype(np.float32)


def _check_costs(hip, oracle, dims, W, cells, start, goal, rows, pad_mode=0, weights=(1.0, 1000.0, 1.0),
                 c_obs=1.0, lam=None):
    N = dims * W
    lam = rows.shape[0] if lam is None else lam
    ctx = hip.HipContext(N, max(lam, 2), 4, dims=dims, waypoints=W, start=start, goal=goal, pad_mode=pad_mode,
                         w_len=weights[0], w_grid=weights[1], w_smooth=weights[2])
    ctx.set_grid(cells, c_obs=c_obs)
    got, gterms = ctx.score_rows(rows)
    co = oracle.CostOracle(dims, W, cells, start, goal, weights=weights, c_obs=c_obs, pad_mode=pad_mode)
    want, wterms = co.cost_rows_f32(rows)
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert rel.max() <= 1e-5, (rel.max(), int(rel.argmax()))
    trel = np.abs(gterms - wterms) / np.maximum(1.0, np.abs(wterms))
    assert trel.max() <= 1e-5
    ctx.close()
    return want


@pytest.mark.parametrize("cfg,count", [(1, 16), (2, 4096)])
def test_cost_2d_occupancy_matches_oracle(hip, oracle, cfg, count):
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(cfg)
    rng = np.random.default_rng(cfg)
    rows = _random_rows(rng, prob, count, prob.sigma0)
    rows[0] = prob.x0.astype(np.float32)                    # the straight line
    rows[1, :] = 0.0                                         # everything in one corner
    rows[2] = np.round(rows[2]) + 0.5                        # exact .5 ties: round-half-even
    want = _check_costs(hip, oracle, 2, prob.W, prob.occ, prob.start, prob.goal, rows)
    assert want.min() > 0 and np.ptp(want) > 1.0


def test_cost_grid_staged_in_lds_equals_grid_in_l2(hip, oracle, monkeypatch):
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    rows = _random_rows(np.random.default_rng(9), prob, 1000, prob.sigma0)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LMCMA_HIP_GRID_LDS", mode)
        ctx = hip.HipContext(prob.N, 4096, 16, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal)
        ctx.set_grid(prob.occ)
        out[mode] = ctx.score_rows(rows)
        ctx.close()
    np.testing.assert_array_equal(out["0"][0], out["1"][0])   # same arithmetic, bit-identical
    np.testing.assert_array_equal(out["0"][1], out["1"][1])


@pytest.mark.parametrize("W", [1, 2, 5, 32, 48, 50, 96, 128])
def test_cost_waypoint_counts_and_pad_modes(hip, oracle, W):
    size = 64
    from lmcma_path_planner_amd import grids
    occ = grids.add_random_boxes(grids.two_bar_map(size), 77, 0.2, [63.0, 0.0], [0.0, 63.0])
    rng = np.random.default_rng(W)
    rows = rng.uniform(0, size - 1, size=(37, 2 * W)).astype(np.float32)
    for pad in (0, 1):
        _check_costs(hip, oracle, 2, W, occ, [63.0, 0.0], [0.0, 63.0], rows, pad_mode=pad,
                     weights=(100.0, 1.0, 0.5), c_obs=3.0, lam=64)


def test_cost_3d_occupancy_and_esdf(hip, oracle):
    from lmcma_path_planner_amd import grids
    size, W = 48, 48
    start, goal = [47.0, 0.0, 0.0], [0.0, 47.0, 47.0]
    occ = grids.add_random_boxes(grids.two_bar_map(size, size), 3, 0.15, start, goal)
    rng = np.random.default_rng(0)
    rows = rng.uniform(0, size - 1, size=(300, 3 * W)).astype(np.float32)
    _check_costs(hip, oracle, 3, W, occ, start, goal, rows, lam=512)
    esdf_cost = grids.state_cost_from_distance(grids.edt(occ), 0.5)
    _check_costs(hip, oracle, 3, W, esdf_cost, start, goal, rows, lam=512)
    # 1-D problem on a float map
    line = np.linspace(0.1, 2.0, 40, dtype=np.float32).reshape(1, 40)
    rows1 = rng.uniform(0, 39, size=(20, 16)).astype(np.float32)
    _check_costs(hip, oracle, 1, 16, line[0], [0.0], [39.0], rows1, lam=32)


@pytest.mark.parametrize("shape,kind", [((100, 100), "two_bar"), ((512, 512), "boxes"), ((48, 48, 48), "boxes"),
                                        ((1, 37), "sparse"), ((33, 70, 129), "sparse"), ((20, 24), "empty"),
                                        ((600, 40), "sparse"), ((128, 128, 128), "boxes")])
def test_device_distance_transform_is_exact(hip, oracle, shape, kind):
    """Squared Euclidean distances are integers, so the device transform must give the oracle's
    distances (exact lower envelopes in fp64) bit for bit, and the installed cost map must score
    rows like a host-built map."""
    from lmcma_path_planner_amd import grids
    rng = np.random.default_rng(17)
    dims = len(shape) if shape[0] > 1 else 1
    start = [float(shape[-1] - 1), 0.0, 0.0][:dims]
    goal = [0.0] + [float(n - 1) for n in shape[::-1][1:]]
    goal = goal[:dims]
    if kind == "two_bar":
        occ = oracle.two_bar_map(shape[0])
    elif kind == "boxes":
        occ = oracle.random_boxes(np.zeros(shape, dtype=np.uint8), 5, 0.15, start, goal)
    elif kind == "sparse":
        occ = (rng.random(shape) < 0.002).astype(np.uint8)
        occ.flat[rng.integers(occ.size)] = 1
    else:
        occ = np.zeros(shape, dtype=np.uint8)
    want = oracle.edt_exact(occ)
    if kind == "empty":
        assert np.all(want == np.float32(1.0e6))
    W = 8
    ctx = hip.HipContext(dims * W, 64, 4, dims=dims, waypoints=W, start=start, goal=goal)
    got = ctx.set_grid_from_occupancy(occ if dims > 1 else occ[0], d_min=0.5, want_distances=True)
    np.testing.assert_array_equal(got.reshape(want.shape), want)
    # scoring against the device-built map equals scoring against the same map built on the host
    rows = np.stack([rng.uniform(0, n - 1, size=(50, W)) for n in shape[::-1][:dims]], axis=1).reshape(50, dims * W)
    rows = rows.astype(np.float32)
    a, ta = ctx.score_rows(rows)
    cost_map = grids.state_cost_from_distance(want, 0.5)
    ctx.set_grid(cost_map if dims > 1 else cost_map[0])
    b, tb = ctx.score_rows(rows)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ta, tb)
    ctx.close()


def test_voxel_file_to_clearance_map_to_planned_path(hip, oracle, tmp_path):
    """The steps before the path, end to end: a binvox voxel model (the format of the reference's
    room model) is read, its clearance map is built on the device, and the device loop plans
    through it; the result is scored by the CPU oracle on a host-built map."""
    from lmcma_path_planner_amd import grids
    size, W = 40, 16
    start, goal = [39.0, 0.0, 0.0], [0.0, 39.0, 39.0]
    occ0 = grids.add_random_boxes(grids.two_bar_map(size, size), 11, 0.12, start, goal)
    path = tmp_path / "model.binvox"
    grids.write_binvox(str(path), occ0, translate=(-1.5, -2.0, -0.001), scale=4.0)
    occ, tr, sc = grids.read_binvox(str(path))
    np.testing.assert_array_equal(occ, occ0)
    N, lam, m = 3 * W, 256, 8
    t = np.arange(1, W + 1) / (W + 1)
    x0 = np.concatenate([start[d] + t * (goal[d] - start[d]) for d in range(3)])
    ctx = hip.HipContext(N, lam, m, dims=3, waypoints=W, start=start, goal=goal, seed=3)
    ctx.set_grid_from_occupancy(occ, d_min=0.5)
    ctx.set_bounds(np.zeros(N), np.full(N, size - 1.0))
    ctx.opt_init(x0, 2.0)
    ctx.run(150)
    st = ctx.opt_get()
    cost_map = oracle.edt_exact(occ)
    cost_map = (1.0 / np.maximum(cost_map.astype(np.float64), 0.5)).astype(np.float32)
    co = oracle.CostOracle(3, W, cost_map, start, goal)
    straight, _ = co.cost(x0)
    final, _ = co.cost(st["xmean"])
    assert st["itr"] == 150 and final < straight
    ctx.close()


def _reference_order(costs):
    c = np.asarray(costs, dtype=np.float32).astype(np.float64)
    c = np.where(np.isnan(c), np.inf, c)
    return np.argsort(c, kind="stable")


@pytest.mark.parametrize("lam", [2, 16, 100, 4096, 5000, 65536, 131072, 1 << 20])   # the last: LMCMA_HIP_MAX_LAMBDA
def test_select_is_exact(hip, oracle, lam):
    rng = np.random.default_rng(lam)
    costs = rng.gamma(2.0, 500.0, size=lam)
    costs[rng.integers(0, lam, size=lam // 4)] = 123.25           # many ties
    if lam >= 16:
        costs[3] = np.inf
        costs[5] = np.nan
        costs[7] = -0.0
        costs[9] = 0.0
        costs[11] = -5.0
    ctx = hip.HipContext(4, lam, 1, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    ctx.set_costs(costs)
    sc, perm = ctx.select()
    order = _reference_order(costs)
    np.testing.assert_array_equal(perm, order)
    want_sorted = np.where(np.isnan(costs), np.inf, costs).astype(np.float32)[order]
    np.testing.assert_array_equal(sc, want_sorted)
    if lam <= 5000:
        s2, i2 = oracle.sort(costs.astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(perm, i2)
    ctx.close()


def test_recombine_matches_fp64_sum(hip, oracle):
    N, lam, m = 144, 1024, 8
    o = oracle.OracleLMCMA(N, lam, m, initial=np.linspace(0, 50, N), sigma=3.0, seed=4)
    ctx = hip.HipContext(N, lam, m, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    rng = np.random.default_rng(1)
    Z = rng.normal(size=(lam, N))
    ctx.set_state(o.xmean, 3.0, None, None, None, 1.0)
    ctx.sample(0, Z)
    rows = ctx.rows()
    costs = rng.random(lam)
    ctx.set_costs(costs)
    _, perm = ctx.select()
    got = ctx.recombine(o.weights)
    want = (o.weights[:, None] * rows[perm[: o.mu]]).sum(axis=0)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)
    ctx.close()


@pytest.mark.parametrize("N,lam,m,gens,dims,W", [(64, 16, 4, 30, 2, 32), (64, 16, 16, 40, 2, 32),
                                                 (128, 256, 16, 24, 2, 64), (36, 8, 3, 25, 3, 12),
                                                 (144, 512, 32, 40, 3, 48),    # 3-D, 48 waypoints, m = 32
                                                 (256, 64, 48, 56, 2, 128),    # 128 waypoints, m = 48
                                                 (33, 32, 5, 12, 3, 11),       # odd parameter count
                                                 (288, 128, 16, 20, 3, 96),    # 96 waypoints in 3-D
                                                 (64, 8192, 8, 12, 2, 32),     # 256 partial rows: two-level sum
                                                 (512, 64, 16, 20, 2, 256),    # longest vectors, two-part update
                                                 (512, 96, 64, 70, 2, 256),    # largest store: vectors stay in global memory
                                                 (320, 32, 24, 30, 2, 160),    # 20 coordinates per lane, m = 24
                                                 (40, 50, 6, 14, 2, 20),       # a population no tile size divides
                                                 (128, 1000, 16, 20, 2, 64),   # the same at the headline's row length
                                                 (3, 7, 2, 8, 3, 1),           # one waypoint, odd population
                                                 (4, 2, 2, 6, 2, 2)])          # smallest population
def test_device_update_follows_oracle_generation_by_generation(hip, oracle, N, lam, m, gens, dims, W):
    """Each generation the oracle is handed the device's rows and costs; its fp64
    update (the reference's order of operations) must then agree with the device's."""
    from lmcma_path_planner_amd import grids
    size = 100
    start = [99.0, 0.0] + ([0.0] if dims == 3 else [])
    goal = [0.0, 99.0] + ([99.0] if dims == 3 else [])
    occ = grids.two_bar_map(size, size if dims == 3 else 1)
    t = np.arange(1, W + 1) / (W + 1)
    x0 = np.concatenate([start[d] + t * (goal[d] - start[d]) for d in range(dims)])
    lo, hi = np.zeros(N), np.full(N, 99.0)
    seed = 42
    o = oracle.OracleLMCMA(N, lam, m, initial=x0, lo=lo, hi=hi, sigma=5.0, seed=seed, z_source=oracle.Z_GIVEN)
    ctx = hip.HipContext(N, lam, m, dims=dims, waypoints=W, start=start, goal=goal, seed=seed)
    ctx.set_grid(occ)
    ctx.set_bounds(lo, hi)
    ctx.opt_init(x0, 5.0)
    for g in range(gens):
        ctx.phase_local()
        rows, costs = ctx.rows(), ctx.costs()
        ctx.phase_rank()
        ctx.phase_update(0)
        o.set_arx(rows)
        o.update_only(costs)
        st = ctx.opt_get()
        vec = ctx.opt_vectors()
        assert st["itr"] == o.itr == g + 1 and st["iterator_sz"] == o.iterator_sz
        np.testing.assert_array_equal(vec["iterator"][: o.iterator_sz], o.iterator[: o.iterator_sz])
        np.testing.assert_allclose(st["xmean"], o.xmean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(st["sigma"], o.sigma, rtol=1e-12)
        np.testing.assert_allclose(st["s"], o.s, rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(vec["pc"], o.pc, rtol=1e-10, atol=1e-12)
        live = o.iterator[: o.iterator_sz]
        np.testing.assert_allclose(vec["pc_arr"][live], o.pc_arr[live], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(vec["v_arr"][live], o.v_arr[live], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(vec["nj"][live], o.nj[live], rtol=1e-9)
        np.testing.assert_allclose(vec["lj"][live], o.lj[live], rtol=1e-9)
        assert st["best_f"] == o.BestF and st["counteval"] == o.counteval
    assert gens > m  # the slot-recycling branch of the direction store was exercised
    ctx.close()


@pytest.mark.parametrize("lam,m", [(512, 16), (8192, 16), (8192, 32), (20480, 8), (1000, 16), (4100, 16), (50, 6)])
def test_graph_run_equals_phase_by_phase_run(hip, lam, m):
    """lambda = 8192 leaves 256 partial rows (summed in two levels), 20480 goes through the
    sorting path; m = 32 uses the one-kernel update, m <= 16 the two-part one."""
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    results = []
    for mode in ("run", "phases"):
        ctx = hip.HipContext(prob.N, lam, m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=7)
        ctx.set_grid(prob.occ)
        ctx.set_bounds(prob.lo, prob.hi)
        ctx.opt_init(prob.x0, prob.sigma0)
        if mode == "run":
            # the first call: one eager generation, then recorded graphs of 32, 8, 2 and 1 generations;
            # the odd calls after it start from either of the two alternating sorted-cost buffers;
            # a call length seen before (3 from the other buffer, then 3 and 12 again) replays the
            # graph of the whole call that was recorded behind its first use
            for n in (12, 1, 3, 11, 3, 3, 12, 3):
                ctx.run(n)
        else:
            for _ in range(48):
                ctx.phase_local()
                ctx.phase_rank()
                ctx.phase_update(0)
        st = ctx.opt_get()
        results.append(st)
        ctx.close()
    a, b = results
    assert a["itr"] == b["itr"] == 48 and a["counteval"] == 48 * lam
    np.testing.assert_array_equal(a["xmean"], b["xmean"])     # deterministic kernels: bit-identical
    assert a["sigma"] == b["sigma"] and a["best_f"] == b["best_f"]


def test_more_call_lengths_than_the_library_keeps_graphs_for(hip):
    """lmcma_hip_run keeps a graph of a whole call for four call lengths; further lengths are cut
    into the recorded pieces every time.  Either way the state equals the phase-by-phase one."""
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    lam, m = 512, 16
    lengths = (3, 3, 5, 5, 7, 7, 9, 9, 11, 11, 13, 13, 3, 13)
    results = []
    for mode in ("run", "phases"):
        ctx = hip.HipContext(prob.N, lam, m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=11)
        ctx.set_grid(prob.occ)
        ctx.set_bounds(prob.lo, prob.hi)
        ctx.opt_init(prob.x0, prob.sigma0)
        if mode == "run":
            for n in lengths:
                ctx.run(n)
        else:
            for _ in range(sum(lengths)):
                ctx.phase_local()
                ctx.phase_rank()
                ctx.phase_update(0)
        results.append(ctx.opt_get())
        ctx.close()
    a, b = results
    assert a["itr"] == b["itr"] == sum(lengths)
    np.testing.assert_array_equal(a["xmean"], b["xmean"])
    assert a["sigma"] == b["sigma"] and a["best_f"] == b["best_f"]


def test_stop_rule_fires_in_the_oracles_generation_and_freezes_the_state(hip, oracle):
    """LMCMA::isBehaviorLearningDone (lmcma.cpp:426-429: sigma < 1e-20) evaluated by the update
    kernel.  Float rows resolve a step size of 1e-20 only around the origin, so the problem is a
    closed path through the origin on an empty map, started at its optimum with sigma = 2e-19: no
    generation improves on the last one, the success rule shrinks sigma, and the oracle (handed the
    device's rows and costs) and the device must cross 1e-20 in the same generation."""
    W, N, lam, m = 8, 16, 32, 4
    start = goal = [0.0, 0.0]
    ctx = hip.HipContext(N, lam, m, dims=2, waypoints=W, start=start, goal=goal, seed=5)
    ctx.set_grid(np.zeros((8, 8), dtype=np.uint8))
    lo, hi = np.full(N, -1.0), np.full(N, 1.0)
    ctx.set_bounds(lo, hi)
    ctx.opt_init(np.zeros(N), 2e-19)
    o = oracle.OracleLMCMA(N, lam, m, initial=np.zeros(N), lo=lo, hi=hi, sigma=2e-19, seed=5, z_source=oracle.Z_GIVEN)
    fired = None
    for g in range(200):
        ctx.phase_local()
        rows, costs = ctx.rows(), ctx.costs()
        assert np.unique(costs).size > lam // 2          # the costs resolve the candidates: no mass of ties
        ctx.phase_rank()
        ctx.phase_update(0)
        o.set_arx(rows)
        o.update_only(costs)
        st = ctx.opt_get()
        np.testing.assert_allclose(st["sigma"], o.sigma, rtol=1e-11)
        assert st["done"] == o.done(), (g, st["sigma"], o.sigma)
        if st["done"]:
            fired = g + 1
            break
    assert fired is not None and fired == st["itr"] and st["sigma"] < 1e-20
    # generations queued beyond the stop change nothing
    vec = ctx.opt_vectors()
    ctx.run(20)
    for _ in range(3):
        ctx.phase_local()
        ctx.phase_rank()
        ctx.phase_update(0)
    st2, vec2 = ctx.opt_get(), ctx.opt_vectors()
    assert st2["done"] and st2["itr"] == fired and st2["sigma"] == st["sigma"] and st2["counteval"] == st["counteval"]
    np.testing.assert_array_equal(st2["xmean"], st["xmean"])
    np.testing.assert_array_equal(vec2["v_arr"], vec["v_arr"])
    # a new query starts live again
    ctx.opt_reset()
    ctx.run(3)
    st3 = ctx.opt_get()
    assert not st3["done"] and st3["itr"] == 3
    ctx.close()
    # the class: step(n) returns at the stop and isBehaviorLearningDone reports it
    from lmcma_path_planner_amd import LMCMA
    opt = LMCMA(np.zeros(N), lam, lo, hi, 2e-19, None, 5)
    opt.setOptions(m=m, philox=True, device_update=True)
    opt.setTrajectoryProblem(2, W, np.zeros((8, 8), dtype=np.uint8), start, goal)
    opt.init(N)
    assert not opt.isBehaviorLearningDone()
    opt.step(5000)
    assert opt.isBehaviorLearningDone() and opt.generation < 200 and opt.counteval == opt.generation * lam
    opt.close()


def _run_both(hip, oracle, N, W, lam, m, occ, start, goal, x0, lo, hi, sigma0, gens, seed):
    co = oracle.CostOracle(2, W, occ, start, goal)
    o = oracle.OracleLMCMA(N, lam, m, initial=x0, lo=lo, hi=hi, sigma=sigma0, seed=seed, z_source=oracle.Z_PHILOX)
    for _ in range(gens):
        c, _t = co.cost_rows_f32(o.arx.astype(np.float32))
        o.tell_all(c)
    ctx = hip.HipContext(N, lam, m, dims=2, waypoints=W, start=start, goal=goal, seed=seed)
    ctx.set_grid(occ)
    ctx.set_bounds(lo, hi)
    ctx.opt_init(x0, sigma0)
    ctx.run(gens)
    st = ctx.opt_get()
    ctx.close()
    return co, co.cost(o.xmean), co.cost(st["xmean"])


def test_end_to_end_converged_path_cost_matches_cpu_oracle(hip, oracle):
    """Same Philox stream, same problem, device (fp32 rows) against the CPU oracle (fp64):
    the cost of the converged path agrees to the stated 1e-4 relative.  Obstacle-free map,
    so the cost is convex with a unique optimum (a straight path) and 'converged' is well
    defined; the runs differ in rounding from the first generation on."""
    W, N, lam, m, gens = 32, 64, 32, 8, 4000
    occ = np.zeros((100, 100), dtype=np.uint8)
    start, goal = [99.0, 0.0], [0.0, 99.0]
    t = np.arange(1, W + 1) / (W + 1)
    line = np.concatenate([start[d] + t * (goal[d] - start[d]) for d in range(2)])
    x0 = line + np.concatenate([6 * np.sin(3 * np.pi * t), 4 * np.sin(2 * np.pi * t)])
    co, (cpu_cost, _), (gpu_cost, _) = _run_both(hip, oracle, N, W, lam, m, occ, start, goal, x0, np.zeros(N),
                                                 np.full(N, 99.0), 1.0, gens, 42)
    even, _ = co.cost(line)   # evenly spaced straight path: near the optimum, which spaces the ends differently
    rel = abs(gpu_cost - cpu_cost) / cpu_cost
    print("converged path cost: gpu %.8f cpu %.8f evenly-spaced line %.8f rel %.3e" % (gpu_cost, cpu_cost, even, rel))
    assert rel <= 1e-4
    assert 140.0 < gpu_cost <= even and 140.0 < cpu_cost <= even


def test_end_to_end_obstacle_map_reaches_a_collision_free_path(hip, oracle):
    """cfg1 (100x100 two-bar map, W=32, lambda=16, m=4).  The occupancy cost is piecewise
    constant, so the two runs (which rank near-tied candidates differently) settle in
    different local optima; what must hold for both is a collision-free path far below the
    straight line's cost.  The relative gap between them is printed, not asserted."""
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(1)
    co, (cpu_cost, cpu_terms), (gpu_cost, gpu_terms) = _run_both(
        hip, oracle, prob.N, prob.W, prob.lam, prob.m, prob.occ, prob.start, prob.goal, prob.x0, prob.lo,
        prob.hi, prob.sigma0, 1500, 42)
    straight, _ = co.cost(prob.x0)
    print("obstacle map: gpu %.3f cpu %.3f straight %.1f rel gap %.3e" %
          (gpu_cost, cpu_cost, straight, abs(gpu_cost - cpu_cost) / cpu_cost))
    assert gpu_terms[1] == 0.0 and cpu_terms[1] == 0.0          # no occupied cell on either path
    assert gpu_cost < 0.05 * straight and cpu_cost < 0.05 * straight


def test_lmcma_class_ask_tell_like_the_reference_example(oracle, golden):
    """The reference's only usage example (example_lmcma.cpp:28-76): 1000 ask/tell
    evaluations of the two-Gaussian toy function from (0,0) with bounds [-2,15]."""
    from lmcma_path_planner_amd import LMCMA
    t = golden["toy_function"]
    lo, hi = [t["bounds"][0]] * 2, [t["bounds"][1]] * 2
    best = []
    for seed in (1, 2, 3):
        opt = LMCMA([0.0, 0.0], -1, lo, hi, inseed=seed)
        opt.init(2)
        ref = oracle.OracleLMCMA(2, 0, 0, initial=np.zeros(2), lo=np.array(lo), hi=np.array(hi), seed=seed)
        assert opt.populationSize == ref.lam == 6
        # the first population comes from the same host RNG stream: equal up to fp32 rounding
        np.testing.assert_allclose(opt.population(), ref.arx, rtol=0, atol=2e-6)
        for _ in range(t["evaluations"]):
            x = opt.getNextParameterVector()
            opt.setEvaluationFeedback([oracle.toy_cost(*x)])
        assert opt.counteval == 1000 and opt.generation == 1000 // 6
        best.append(opt.BestF)
        assert min(abs(opt.BestF - t["global_min_value_approx"]), abs(opt.BestF - t["local_min_value_approx"])) < 2e-3
        opt.close()
    assert min(best) < -3.99


def test_lmcma_class_sphere_trace_tracks_recorded_reference_run(oracle, golden):
    from lmcma_path_planner_amd import LMCMA
    t = golden["sphere_trace"]
    opt = LMCMA(np.ones(t["N"]), t["lambda"], None, None, t["sigma"], None, t["seed"])
    opt.init(t["N"])
    x = opt.getNextParameterVector()
    assert abs(x[0] - t["first_candidate_first2"][0]) < 1e-6 and abs(x[1] - t["first_candidate_first2"][1]) < 1e-6
    for i in range(t["evaluations"]):
        x = opt.getNextParameterVector()
        opt.setEvaluationFeedback([float(np.sum(x * x))])
    # fp32 rows instead of fp64 ones perturb the run; the reference's BestF is 7.45e-4
    assert 1e-5 < opt.BestF < 2e-2
    assert opt.generation == 50 and 1e-3 < opt.sigma < 0.5
    opt.close()


def test_lmcma_class_batched_step_with_host_update_and_structured_covariance(oracle):
    from lmcma_path_planner_amd import LMCMA, grids, planner
    prob = grids.benchmark_problem(1)
    co = oracle.CostOracle(2, prob.W, prob.occ, prob.start, prob.goal)
    straight, _ = co.cost(prob.x0)
    for use_cov in (False, True):
        cov = planner.covariance(2, prob.W) if use_cov else None
        opt = LMCMA(prob.x0, prob.lam, prob.lo, prob.hi, prob.sigma0 * (8.0 if use_cov else 1.0), cov, 42)
        opt.setOptions(m=prob.m)
        opt.setTrajectoryProblem(2, prob.W, prob.occ, prob.start, prob.goal)
        opt.init(prob.N)
        pop = opt.population()
        opt.step(1)
        # costs fed back by the device equal the oracle's on the same rows
        want, _ = co.cost_rows_f32(pop.astype(np.float32))
        got = opt.lastCosts()
        assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-5
        opt.step(400)
        final, _ = co.cost(opt.mean())
        assert opt.counteval == 401 * prob.lam and opt.BestF < straight and final < straight
        opt.close()


@pytest.mark.parametrize("use_cov", [False, True])
def test_lmcma_class_device_update_mode(oracle, use_cov):
    from lmcma_path_planner_amd import LMCMA, grids, planner
    prob = grids.benchmark_problem(1)
    co = oracle.CostOracle(2, prob.W, prob.occ, prob.start, prob.goal)
    # the constructor's covariance argument reaches the device sampler as its Cholesky factor
    cov = planner.covariance(2, prob.W) if use_cov else None
    opt = LMCMA(prob.x0, prob.lam, prob.lo, prob.hi, prob.sigma0 * (8.0 if use_cov else 1.0), cov, 42)
    opt.setOptions(m=prob.m, philox=True, device_update=True)
    opt.setTrajectoryProblem(2, prob.W, prob.occ, prob.start, prob.goal)
    opt.init(prob.N)
    opt.step(300)
    straight, _ = co.cost(prob.x0)
    final, _ = co.cost(opt.mean())
    assert opt.generation == 300 and opt.counteval == 300 * prob.lam
    assert final < straight and opt.BestF < straight
    opt.close()


def test_lmcma_class_ask_tell_with_the_device_update_follows_the_oracle(oracle):
    """The reference's calling convention (getNextParameterVector / setEvaluationFeedback with the
    caller's own cost, example_lmcma.cpp:47-54) with the state on the device: candidates drawn
    there from the Philox stream, the caller's costs sent back, ranking, recombination and the
    fp64 update on the device.  The oracle is handed the device's rows and the costs as floats
    (the device ranks floats) and must agree generation by generation."""
    from lmcma_path_planner_amd import LMCMA
    N, lam, m, seed = 24, 64, 8, 11
    x0, lo, hi = np.full(N, 3.0), np.full(N, -4.0), np.full(N, 6.0)
    scale = 10.0 ** (np.arange(N) / (N - 1))
    opt = LMCMA(x0, lam, lo, hi, 1.5, None, seed)
    opt.setOptions(m=m, philox=True, device_update=True)
    opt.init(N)
    ref = oracle.OracleLMCMA(N, lam, m, initial=x0, lo=lo, hi=hi, sigma=1.5, seed=seed, z_source=oracle.Z_GIVEN)
    best = np.inf
    for g in range(m + 14):
        rows, costs = [], []
        for i in range(lam):
            x = opt.getNextParameterVector()
            f = float(np.sum(scale * (x - 1.0) ** 2))
            opt.setEvaluationFeedback([0.5 * f, 0.5 * f])
            rows.append(x)
            costs.append(f)
        rows, costs = np.array(rows), np.array(costs)
        best = min(best, costs.min())
        if g == 0:
            z = np.stack([oracle.philox_normals(seed, 0, i, N) for i in range(lam)])
            assert np.max(np.abs(rows - np.clip(x0 + 1.5 * z, lo, hi))) <= 2e-5 * 10.0
        ref.set_arx(rows)
        ref.update_only(costs.astype(np.float32).astype(np.float64))
        np.testing.assert_allclose(opt.mean(), ref.xmean, rtol=1e-12, atol=1e-12, err_msg="generation %d" % g)
        np.testing.assert_allclose(opt.sigma, ref.sigma, rtol=1e-12)
        assert opt.generation == g + 1 and opt.counteval == (g + 1) * lam
    assert opt.BestF == best     # kept on the host from the caller's doubles (lmcma.cpp:192-198)
    assert best < 0.2 * float(np.sum(scale * 4.0))
    opt.close()


def test_sepcma_class_follows_the_oracle_generation_by_generation(oracle):
    """SepCMA behind the reference's interface: device sampling from the reference's RNG stream
    (fp32 rows), host update; the oracle is handed the device's rows each generation."""
    from lmcma_path_planner_amd import SepCMA
    n, lam, seed = 24, 16, 9
    lo, hi = np.full(n, -4.0), np.full(n, 6.0)
    scale = 10.0 ** (np.arange(n) / (n - 1))
    opt = SepCMA(np.full(n, 3.0), lam, lo, hi, 1.5, None, seed)
    opt.init(n)
    ref = oracle.OracleSepCMA(n, lam, initial=np.full(n, 3.0), lo=lo, hi=hi, sigma=1.5, seed=seed)
    for g in range(60):
        rows = opt.population()
        assert np.max(np.abs(rows - ref.arx)) <= 2e-6 * (np.abs(ref.arx).max() + 1.0), g
        costs = np.sum(scale * (rows - 1.0) ** 2, axis=1)
        for i in range(lam):
            x = opt.getNextParameterVector()
            np.testing.assert_array_equal(x, rows[i])
            opt.setEvaluationFeedback([costs[i]])
        ref.set_arx(rows)
        ref.tell_all(costs)
        np.testing.assert_allclose(opt.mean(), ref.xmean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(opt.sigma, ref.sigma, rtol=1e-12)
        np.testing.assert_allclose(opt.diagonal(), ref.diag, rtol=1e-12)
    assert opt.generation == 60 and opt.counteval == 60 * lam
    assert opt.BestF < 1e-2 * np.sum(scale * 4.0)
    opt.close()


def test_sepcma_class_plans_on_the_device_cost(oracle):
    from lmcma_path_planner_amd import SepCMA, grids
    prob = grids.benchmark_problem(1)
    co = oracle.CostOracle(2, prob.W, prob.occ, prob.start, prob.goal)
    straight, _ = co.cost(prob.x0)
    opt = SepCMA(prob.x0, 32, prob.lo, prob.hi, prob.sigma0, None, 42)
    opt.setTrajectoryProblem(2, prob.W, prob.occ, prob.start, prob.goal)
    opt.init(prob.N)
    opt.step(300)
    final, _ = co.cost(opt.mean())
    assert opt.counteval == 300 * 32 and final < straight and opt.BestF < straight
    opt.close()


def test_cmachol_class_follows_the_oracle_generation_by_generation(oracle):
    """CMAChol behind the reference's interface: normals from the reference's RNG stream, the dense
    product A z on the device in fp32, the factor update on the host; the oracle is handed the
    device's rows each generation."""
    from lmcma_path_planner_amd import CMAChol
    n, lam, seed = 40, 24, 4
    lo, hi = np.full(n, -5.0), np.full(n, 7.0)
    rng = np.random.default_rng(6)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    H = Q @ np.diag(10.0 ** np.linspace(0, 2, n)) @ Q.T
    opt = CMAChol(np.full(n, 3.0), lam, lo, hi, 1.5, None, seed)
    opt.init(n)
    ref = oracle.OracleCMAChol(n, lam, initial=np.full(n, 3.0), lo=lo, hi=hi, sigma=1.5, seed=seed)
    for g in range(80):
        rows = opt.population()
        scale = np.abs(ref.xmean).max() + np.abs(ref.arx - ref.xmean).max() + 1.0
        assert np.max(np.abs(rows - ref.arx)) <= 1e-5 * scale, g
        X = rows - 1.0
        costs = np.einsum("ij,jk,ik->i", X, H, X)
        for i in range(lam):
            opt.getNextParameterVector()
            opt.setEvaluationFeedback([costs[i]])
        ref.set_arx(rows)
        ref.tell_all(costs)
        np.testing.assert_allclose(opt.mean(), ref.xmean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(opt.sigma, ref.sigma, rtol=1e-12)
        np.testing.assert_allclose(opt.factor(), ref.factor, rtol=1e-10, atol=1e-12)
    assert opt.generation == 80 and opt.BestF < 0.1 * float(np.full(n, 2.0) @ H @ np.full(n, 2.0))
    opt.close()


def test_error_reporting(hip):
    from lmcma_path_planner_amd._lib import LmcmaError
    with pytest.raises(LmcmaError, match="n_params"):
        hip.HipContext(4096, 16, 4)
    with pytest.raises(LmcmaError, match="dims"):
        hip.HipContext(64, 16, 4, dims=2, waypoints=31)
    ctx = hip.HipContext(64, 16, 4, dims=2, waypoints=32)
    with pytest.raises(LmcmaError, match="set_grid"):
        ctx.opt_init(np.zeros(64), 1.0)
        ctx.run(1)
    with pytest.raises(LmcmaError, match="nz must be 1"):
        ctx.set_grid(np.zeros((2, 4, 4), dtype=np.uint8))
    with pytest.raises(LmcmaError, match="nz must be 1"):
        ctx.set_grid_from_occupancy(np.zeros((2, 4, 4), dtype=np.uint8))
    with pytest.raises(LmcmaError, match="d_min"):
        ctx.set_grid_from_occupancy(np.zeros((4, 4), dtype=np.uint8), d_min=0.0)
    with pytest.raises(LmcmaError, match="1024"):
        ctx.set_grid_from_occupancy(np.zeros((2, 1025), dtype=np.uint8))
    with pytest.raises(ValueError):
        ctx.set_cov_factor(np.eye(5))
    with pytest.raises(LmcmaError, match="not finite"):
        ctx.set_cov_factor(np.full((64, 64), np.nan))
    assert 0.0 < ctx.timing_event_overhead(64) < 1.0   # milliseconds per empty event bracket
    ctx.close()
    # a factor also applies to normals the caller hands in
    rng = np.random.default_rng(2)
    A = rng.normal(size=(64, 64)) / 8.0
    Z = rng.normal(size=(16, 64))
    ctx = hip.HipContext(64, 16, 4, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    ctx.set_cov_factor(np.ascontiguousarray(A.T))            # L[c*N + r] = A(r, c)
    ctx.set_state(np.zeros(64), 1.0, None, None, None, 1.0)
    ctx.sample(0, Z)
    want = Z.astype(np.float32).astype(np.float64) @ A.T
    assert np.max(np.abs(ctx.rows() - want)) <= 1e-5 * np.abs(want).max()
    ctx.set_cov_factor(None)
    ctx.sample(0, Z)
    assert np.max(np.abs(ctx.rows() - Z)) <= 1e-6
    ctx.close()


def test_documented_limits_are_the_limits(hip):
    """The sizes include/lmcma_hip.h states as maxima are accepted and work; one beyond each is refused
    with an error, not a fault."""
    from lmcma_path_planner_amd._lib import LmcmaError
    tiny = dict(z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    with pytest.raises(LmcmaError, match="lambda_total"):
        hip.HipContext(4, (1 << 20) + 1, 1, **tiny)
    with pytest.raises(LmcmaError, match="lambda_total"):
        hip.HipContext(4, 1, 1, **tiny)
    with pytest.raises(LmcmaError, match="n_params"):
        hip.HipContext(513, 16, 4, **tiny)
    with pytest.raises(LmcmaError, match="m must be"):
        hip.HipContext(8, 16, 4097, **tiny)
    with pytest.raises(LmcmaError, match="queries"):
        hip.HipContext(8, 16, 4, dims=2, waypoints=4, queries=257)
    # the device-resident update keeps at most 64 direction vectors; the stepwise calls take more
    c = hip.HipContext(8, 16, 65, dims=2, waypoints=4)
    with pytest.raises(LmcmaError, match="64 direction vectors"):
        c.opt_init(np.zeros(8), 1.0)
    c.close()
    # 512 parameters, 256 queries of the smallest population the batch path takes, all in one context
    W = 256
    start, goal = [15.0, 0.0], [0.0, 15.0]
    c = hip.HipContext(2 * W, 64, 4, dims=2, waypoints=W, start=start, goal=goal, queries=256, seed=9)
    c.set_grid(np.zeros((16, 16), dtype=np.uint8))
    c.set_bounds(np.zeros(2 * W), np.full(2 * W, 15.0))
    t = np.arange(1, W + 1) / (W + 1)
    c.opt_init(np.concatenate([start[d] + t * (goal[d] - start[d]) for d in range(2)]), 0.5)
    c.run(3)
    for q in (0, 255):
        c.query_select(q)
        st = c.opt_get()
        assert st["itr"] == 3 and np.isfinite(st["best_f"]) and np.all(np.isfinite(st["xmean"]))
    c.close()


def test_two_shards_on_one_gpu_match_the_unsharded_population(hip):
    """The sharded phases with the exchange steps done by hand (device-to-device copies
    through torch views of the library's buffers, as the RCCL path aliases them)."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    lam, half, gens = 1024, 512, 12
    dev = torch.device("cuda", 0)

    def make(first, nloc):
        ctx = hip.HipContext(prob.N, lam, prob.m, dims=2, waypoints=prob.W, lambda_local=nloc, first_candidate=first,
                             start=prob.start, goal=prob.goal, seed=99)
        ctx.set_grid(prob.occ)
        ctx.set_bounds(prob.lo, prob.hi)
        ctx.opt_init(prob.x0, prob.sigma0)
        return ctx

    whole = make(0, lam)
    parts = [make(0, half), make(half, half)]
    shards = [shard_mod.HipShard(c, dev) for c in parts]
    N = prob.N
    for _ in range(gens):
        whole.phase_local()
        whole.phase_rank()
        whole.phase_update(0)
        for c in parts:
            c.phase_local()
        for c in parts:
            c.synchronize()
        # "all-gather" of the costs: each shard's block into the other's full vector
        shards[0].cost_all[half:].copy_(shards[1].cost_local)
        shards[1].cost_all[:half].copy_(shards[0].cost_local)
        torch.cuda.synchronize()
        for c in parts:
            c.phase_rank()
        for c in parts:
            c.synchronize()
        for s in shards:
            L = s.partial_len                      # N + 2: elite sum, success-rule count, lowest cost
            s.partial_all[:L].copy_(shards[0].partial_local)
            s.partial_all[L:2 * L].copy_(shards[1].partial_local)
        torch.cuda.synchronize()
        for c in parts:
            c.phase_update(2)
    w = whole.opt_get()
    a, b = parts[0].opt_get(), parts[1].opt_get()
    np.testing.assert_array_equal(a["xmean"], b["xmean"])              # replicas stay bit-identical
    assert a["sigma"] == b["sigma"] and a["best_f"] == b["best_f"] == w["best_f"]
    np.testing.assert_allclose(a["xmean"], w["xmean"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(a["sigma"], w["sigma"], rtol=1e-11)
    # the last generation's costs: the two halves are the unsharded run's costs
    np.testing.assert_allclose(np.concatenate([parts[0].costs(), parts[1].costs()]), whole.costs(), rtol=1e-5)
    for c in parts + [whole]:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("G,mode,nloc", [(4, "ties", 256), (8, "random", 256), (3, "random", 256),
                                         (8, "random", 4096),     # the 8-GPU job of bench.py: 32768 keys, one LDS chunk
                                         (8, "random", 8192),     # 65536 keys: two chunks of four lists
                                         (4, "random", 16384),    # 65536 keys: lists as long as half a chunk
                                         (2, "ties", 4096)])
def test_merged_ranks_of_shards_equal_the_stable_sort_of_the_whole_population(hip, G, mode, nloc):
    """G shards on one GPU; after the hand-made exchange of the sorted local costs every
    shard's entries of the global permutation and sorted list must be those of a stable
    sort of all costs in candidate order.  "ties": degenerate bounds make every candidate
    the same path, so all costs are equal and the order must be the candidate order, across
    shards too."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    lam = G * nloc
    dev = torch.device("cuda", 0)
    lo, hi = (prob.x0, prob.x0) if mode == "ties" else (prob.lo, prob.hi)
    ctxs = []
    for g in range(G):
        c = hip.HipContext(prob.N, lam, prob.m, dims=2, waypoints=prob.W, lambda_local=nloc, first_candidate=g * nloc,
                           start=prob.start, goal=prob.goal, seed=5)
        c.set_grid(prob.occ)
        c.set_bounds(lo, hi)
        c.opt_init(prob.x0, prob.sigma0)
        ctxs.append(c)
    shards = [shard_mod.HipShard(c, dev) for c in ctxs]
    for gen in range(3):
        for c in ctxs:
            c.phase_local()
        for c in ctxs:
            c.synchronize()
        raw = np.concatenate([c.costs() for c in ctxs]).astype(np.float32)
        for s in shards:                                   # the "all-gather"
            for g, src in enumerate(shards):
                s.cost_all[g * nloc:(g + 1) * nloc].copy_(src.cost_local)
        torch.cuda.synchronize()
        for c in ctxs:
            c.phase_rank()
        for c in ctxs:
            c.synchronize()
        order = _reference_order(raw)
        want_rank = np.empty(lam, dtype=np.int64)
        want_rank[order] = np.arange(lam)
        for g, c in enumerate(ctxs):
            perm = shard_mod.alias_tensor(c, hip.BUF_PERM, torch.int32, dev).cpu().numpy()
            sc = shard_mod.alias_tensor(c, hip.BUF_SORTED_COST, torch.float32, dev).cpu().numpy()
            own = np.arange(g * nloc, (g + 1) * nloc)
            np.testing.assert_array_equal(perm[want_rank[own]], own)          # each own candidate sits at its rank
            np.testing.assert_array_equal(sc[want_rank[own]], raw[own])
        if mode == "ties":
            assert np.all(raw == raw[0])
        L = shards[0].partial_len
        for s in shards:
            for g, src in enumerate(shards):
                s.partial_all[g * L:(g + 1) * L].copy_(src.partial_local)
        torch.cuda.synchronize()
        for c in ctxs:
            c.phase_update(G)
    states = [c.opt_get() for c in ctxs]
    for st in states[1:]:
        np.testing.assert_array_equal(st["xmean"], states[0]["xmean"])       # replicas stay bit-identical
        assert st["sigma"] == states[0]["sigma"] and st["best_f"] == states[0]["best_f"]
    assert states[0]["best_f"] == float(raw.min()) or states[0]["best_f"] <= float(raw.min())
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("lam,mode", [(8192, "random"), (12288, "random"), (12288, "ties"),
                                      (16384, "random"), (16384, "ties"), (20480, "random"), (32768, "random")])
def test_tiled_rank_of_one_context_equals_the_stable_sort(hip, monkeypatch, lam, mode):
    """A whole population above 8192 candidates in one context is ranked in tiles of at most 4096
    (each among itself, by counting) whose ascending lists are then merged.  The permutation and
    the sorted costs must be those of a stable sort of all costs ("ties": degenerate bounds make
    every cost equal, so the order is the candidate order), and the state must follow a context
    that ranks the whole population at once (success-rule count, lowest cost: exactly; the mean:
    another order of the same fp64 sum)."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(2)
    dev = torch.device("cuda", 0)
    lo, hi = (prob.x0, prob.x0) if mode == "ties" else (prob.lo, prob.hi)

    def make():
        c = hip.HipContext(prob.N, lam, 8, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=3)
        c.set_grid(prob.occ)
        c.set_bounds(lo, hi)
        c.opt_init(prob.x0, prob.sigma0)
        return c

    tiled = make()
    monkeypatch.setenv("LMCMA_HIP_NO_TILED_RANK", "1")
    whole = make()
    monkeypatch.delenv("LMCMA_HIP_NO_TILED_RANK")
    for gen in range(4):
        for c in (tiled, whole):
            c.phase_local()
        raw = tiled.costs().astype(np.float32)
        np.testing.assert_array_equal(raw, whole.costs().astype(np.float32))
        for c in (tiled, whole):
            c.phase_rank()
            c.synchronize()
        order = _reference_order(raw)
        for c in (tiled, whole):
            perm = shard_mod.alias_tensor(c, hip.BUF_PERM, torch.int32, dev).cpu().numpy()
            sc = shard_mod.alias_tensor(c, hip.BUF_SORTED_COST, torch.float32, dev).cpu().numpy()
            np.testing.assert_array_equal(perm, order)
            np.testing.assert_array_equal(sc, raw[order])
        for c in (tiled, whole):
            c.phase_update(0)
        a, b = tiled.opt_get(), whole.opt_get()
        assert a["sigma"] == b["sigma"] and a["s"] == b["s"] and a["best_f"] == b["best_f"] <= float(raw.min())
        np.testing.assert_allclose(a["xmean"], b["xmean"], rtol=1e-12, atol=1e-12)
        if mode == "ties":
            assert np.all(raw == raw[0])
    # the replayed loop takes the same path
    tiled.opt_reset()
    whole.opt_reset()
    tiled.run(11)
    whole.run(11)
    a, b = tiled.opt_get(), whole.opt_get()
    assert a["itr"] == b["itr"] == 11
    np.testing.assert_allclose(a["xmean"], b["xmean"], rtol=1e-9, atol=1e-9)
    tiled.close()
    whole.close()


def test_sharded_generation_over_two_processes():
    """dist.sharded_generation as bench.py --gpus 2 runs it, two ranks on this one GPU (gloo: RCCL
    refuses two ranks on one device), against the unsharded population."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(here, "gpu_shard_worker.py"), "12"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
    assert out.returncode == 0 and "sharded vs whole: equal" in out.stdout, out.stdout[-2000:]


def test_cpp_examples_against_the_drop_in_header():
    """examples/*.cpp are written against include/lmcma.hpp like a caller of the reference's
    header; the binaries are built on the CPU box by __graft_entry__.build()."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    toy = os.path.join(root, "examples", "bin", "toy_ask_tell")
    plan = os.path.join(root, "examples", "bin", "plan_path")
    assert os.path.exists(toy) and os.path.exists(plan), "run __graft_entry__.build() first"
    out = subprocess.run([toy, "3", "1000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    best = float(re.search(r"best (-?[0-9.]+)", out.stdout).group(1))
    assert min(abs(best + 4.0007), abs(best + 2.0013)) < 2e-3
    out = subprocess.run([plan, "100", "32", "64", "8", "400"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "generations 400 evaluations 25600" in out.stdout
    assert float(re.search(r"best cost ([0-9.]+)", out.stdout).group(1)) < 2000.0   # straight line: 25596
    sib = os.path.join(root, "examples", "bin", "sibling_optimisers")
    assert os.path.exists(sib), "run __graft_entry__.build() first"
    out = subprocess.run([sib, "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_stepwise_api_with_the_reference_choice_m_equal_lambda(hip, oracle):
    """m = lambda = 256 direction vectors (the reference's only setting) through the stepwise API."""
    N, lam = 64, 256
    o = _evolved_oracle(oracle, N, lam, lam, 40)
    st = _state_of(o)
    assert st["k"] == 40
    Z = np.random.default_rng(3).normal(size=(lam, N))
    o.sample_given(Z.astype(np.float32).astype(np.float64))
    ctx = hip.HipContext(N, lam, lam, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
    ctx.set_state(st["xmean"], st["sigma"], st["V"], st["PC"], st["nj"], st["M"])
    ctx.sample(40, Z)
    scale = np.abs(st["xmean"]).max() + np.abs(o.arx - st["xmean"]).max() + 1.0
    assert np.max(np.abs(ctx.rows() - o.arx)) <= 3e-6 * scale
    from lmcma_path_planner_amd._lib import LmcmaError
    with pytest.raises(LmcmaError, match="at most 64"):
        ctx.opt_init(np.zeros(N), 1.0)
    ctx.close()


GIVE_UP_SCRIPT = """
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %r)
from lmcma_path_planner_amd import _lib, grids, hip
prob = grids.benchmark_problem(2)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=42)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
ctx.run(6)                                   # recorded graphs, the carried update among them: all well
good = ctx.opt_get()
vec = ctx.opt_vectors()
lib = _lib.load()
lib.lmcma_hip_diag_skip_report.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.lmcma_hip_diag_skip_report(ctx._h, 1) == 0
ctx.run(4)                                   # one workgroup withholds its report: the storing one gives up
for call in (ctx.synchronize, lambda: ctx.run(1)):     # a caller that never reads the state sees it too
    try:
        call()
        print("NO ERROR REPORTED")
        sys.exit(1)
    except _lib.LmcmaError as e:
        assert "gave up" in str(e), str(e)
try:
    ctx.opt_get()
    print("NO ERROR REPORTED")
    sys.exit(1)
except _lib.LmcmaError as e:
    assert "gave up" in str(e), str(e)
# what the failed update was to replace is intact: mean, step size, path, stored paths, slot tables of
# generation 6 (the first part of the update, which ran with the ranking launch before, had already
# rebuilt some vectors v_j and their nj, lj from the stored paths; they are re-derived by every update)
sm = _lib.OptSummary()
xm = np.empty(prob.N)
rc = lib.lmcma_hip_opt_get(ctx._h, ctypes.byref(sm), xm.ctypes.data_as(_lib.c_double_p))
assert rc != 0
assert np.array_equal(xm, good["xmean"]), "the mean changed"
after = ctx.opt_vectors()
for k in ("pc", "pc_arr", "iterator"):
    assert np.array_equal(vec[k], after[k]), k
# a reset clears the mark and the context works again
assert lib.lmcma_hip_diag_skip_report(ctx._h, 0) == 0
ctx.opt_reset()
ctx.run(5)
ctx.synchronize()
st = ctx.opt_get()
assert st["itr"] == 5 and np.isfinite(st["best_f"])
print("give-up path: error reported, state intact, reset recovers")
"""


@pytest.mark.gpu
def test_fused_update_give_up_path_reports_and_keeps_the_state():
    """The sampling launch that carries the previous generation's update: its storing workgroup waits
    (bounded) for every other workgroup's report before it overwrites the state.  In the diagnostic
    build one workgroup can be made to withhold its report: the wait gives up, nothing is stored,
    lmcma_hip_synchronize, the next lmcma_hip_run and lmcma_hip_opt_get return the error, what that update was to replace (mean, step size, paths, slot
    tables of the last good generation) is still there, and lmcma_hip_opt_reset puts the context back
    to work."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "lmcma_path_planner_amd", "lib", "liblmcma_hip_diag.so")
    if not os.path.exists(diag):
        pytest.skip("the diagnostic build (make -C lmcma_path_planner_amd/csrc diag) is not there")
    env = dict(os.environ, LMCMA_HIP_LIB=diag)
    out = subprocess.run([sys.executable, "-c", GIVE_UP_SCRIPT % root], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "state intact" in out.stdout
