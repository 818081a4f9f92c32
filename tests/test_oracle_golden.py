"""The CPU oracle against the reference outputs that exist (tests/golden) and
against closed-form properties of the reference algorithm.  CPU only."""
import numpy as np


def test_rng_matches_recorded_reference_values(oracle, golden):
    g = oracle.gauss_stream(7, 3)
    # printed right-to-left (see the fixture's note): reverse to get draw order
    want = list(reversed(golden["rng_gauss_seed7_first3_as_printed"]))
    assert [float(v) for v in g] == want  # bit-exact


def test_rng_uniform_is_minimal_standard_lehmer(oracle):
    # first table entry after the 40 warm-up steps, computed independently
    s = 12345
    table = [0] * 32
    for i in range(39, -1, -1):
        s = (16807 * s) % 2147483647
        if i < 32:
            table[i] = s
    u = oracle.uniform_stream(12345, 1)[0]
    s_next = (16807 * s) % 2147483647
    idx = table[0] // 67108865
    assert u == table[idx] / 2.147483647e9
    assert 0.0 < u < 1.0 and s_next > 0


def test_sphere_trace_matches_recorded_reference_run(oracle, golden):
    t = golden["sphere_trace"]
    o = oracle.OracleLMCMA(t["N"], t["lambda"], 0, initial=np.full(t["N"], t["initial_params_inferred"]),
                           sigma=t["sigma"], seed=t["seed"])
    first = None
    for _ in range(t["evaluations"]):
        x = o.ask()
        if first is None:
            first = x.copy()
        o.tell([float(np.sum(x * x))])
    assert float(first[0]) == t["first_candidate_first2"][0]
    # the survey printed 17 significant digits; equality of the parsed doubles is bit equality
    assert abs(float(first[1]) - t["first_candidate_first2"][1]) <= 2e-17
    assert o.BestF == t["BestF_after_300"]  # 50 generations of update()+sample(), bit-exact
    assert o.counteval == 300 and o.itr == 50


def test_toy_function_optimum(oracle, golden):
    t = golden["toy_function"]
    lo, hi = np.full(2, t["bounds"][0]), np.full(2, t["bounds"][1])
    hits = 0
    for seed in (1, 2, 3, 4, 5):
        o = oracle.OracleLMCMA(2, 0, 0, initial=np.array(t["start"]), lo=lo, hi=hi, seed=seed)
        assert o.lam == 6  # 4 + int(3 ln 2)
        for _ in range(t["evaluations"]):
            x = o.ask()
            o.tell([oracle.toy_cost(*x)])
        if abs(o.BestF - t["global_min_value_approx"]) < 2e-3:
            hits += 1
        else:  # the only other basin is the local minimum near (2, 2)
            assert abs(o.BestF - t["local_min_value_approx"]) < 2e-3
    assert hits >= 1
    assert abs(oracle.toy_cost(4, 4) - (-4.0007)) < 1e-3 and abs(oracle.toy_cost(2, 2) - (-2.0013)) < 1e-3


def test_constants_follow_reference_formulas(oracle):
    for N, lam in ((2, 6), (8, 6), (64, 16), (128, 64)):
        o = oracle.OracleLMCMA(N, lam, 0, initial=np.zeros(N), seed=1)
        mu = lam // 2
        w = np.log(mu + 0.5) - np.log(1 + np.arange(mu))
        w /= w.sum()
        assert o.mu == mu and o.m == lam
        np.testing.assert_allclose(o.weights, w, rtol=1e-15)
        np.testing.assert_allclose(o.mueff, 1.0 / np.sum(w * w), rtol=1e-14)
        assert o.c1 == 1.0 / (10 * np.log(N + 1)) and o.cc == 1.0 / lam


def test_sort_is_stable_and_nan_last(oracle):
    v = np.array([3.0, 1.0, 2.0, 1.0, np.nan, 1.0, -0.0, 0.0])
    s, idx = oracle.sort(v)
    assert list(idx[:2]) == [6, 7]            # -0 == +0: input order kept
    assert list(idx[2:5]) == [1, 3, 5]        # equal values keep candidate order
    assert idx[-1] == 4 and np.isinf(s[-1])   # NaN ranks last


def test_differentiation_matrix_and_covariance(oracle):
    A = oracle.differentiation_matrix(8, 2)
    row = np.array([-1 / 12, 16 / 12, -30 / 12, 16 / 12, -1 / 12])
    np.testing.assert_array_equal(A[3, 1:6], row)
    np.testing.assert_array_equal(A[0, :3], row[2:])          # truncated at the start
    np.testing.assert_array_equal(A[7, 5:], row[:3])          # and at the end
    assert np.count_nonzero(A) == 8 * 5 - 2 * (1 + 2)
    for D, W in ((2, 1), (1, 8), (2, 32)):
        C = oracle.covariance(D, W)
        n = D * W
        Ablk = np.zeros((n, n))
        for d in range(D):
            Ablk[d * W:(d + 1) * W, d * W:(d + 1) * W] = oracle.differentiation_matrix(W, 2)
        R = Ablk @ Ablk
        Cinv_scaled = np.linalg.inv(R)
        Cinv_scaled /= Cinv_scaled.diagonal().max() * W
        np.testing.assert_allclose(C, Cinv_scaled, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(C.diagonal().max(), 1.0 / W, rtol=1e-12)
        L = oracle.cholesky(C)             # column-major lower factor
        np.testing.assert_allclose(L.T @ L.T.T, C, rtol=1e-10, atol=1e-14)
        z = np.arange(1, n + 1, dtype=float)
        np.testing.assert_allclose(oracle.apply_cov_l(L, z), L.T @ z, rtol=1e-12)


def test_m_equal_lambda_is_default_and_decoupled_m_runs(oracle):
    N, lam = 16, 8
    a = oracle.OracleLMCMA(N, lam, 0, initial=np.ones(N), seed=5)
    b = oracle.OracleLMCMA(N, lam, lam, initial=np.ones(N), seed=5)
    c = oracle.OracleLMCMA(N, lam, 3, initial=np.ones(N), seed=5)
    for _ in range(lam * 12):
        for o in (a, b, c):
            x = o.ask()
            o.tell([float(np.sum(x * x))])
    assert a.BestF == b.BestF and a.sigma == b.sigma
    assert c.iterator_sz == 3 and np.isfinite(c.BestF)


def test_philox_known_answer(oracle):
    # Random123 known-answer vectors for philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    z = np.concatenate([oracle.philox_normals(42, g, c, 128) for g in range(4) for c in range(64)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02


def test_cost_contract_on_hand_computed_path(oracle):
    # 2-D, 3 waypoints on the straight line (0,0)->(4,0): only length contributes
    occ = np.zeros((8, 8), dtype=np.uint8)
    co = oracle.CostOracle(2, 3, occ, start=(0, 0), goal=(4, 0), weights=(1.0, 1000.0, 1.0))
    row = np.array([1.0, 2.0, 3.0, 0.0, 0.0, 0.0])
    c, terms = co.cost(row)
    # endpoint padding repeats start/goal two nodes out, so the first and last waypoint of a
    # uniform straight line each see an acceleration of -1/12
    np.testing.assert_allclose(terms, [4.0, 0.0, 2.0 / 144.0], atol=1e-12)
    assert abs(c - (4.0 + 2.0 / 144.0)) < 1e-12
    # occupy the cell of the middle waypoint: trapezoid gives 1/2*(0+1)*1 + 1/2*(1+0)*1 = 1
    occ[0, 2] = 1  # row = y = 0, col = x = 2
    co = oracle.CostOracle(2, 3, occ, start=(0, 0), goal=(4, 0))
    c, terms = co.cost(row)
    np.testing.assert_allclose(terms, [4.0, 1.0, 2.0 / 144.0], atol=1e-12)
    assert abs(c - (1004.0 + 2.0 / 144.0)) < 1e-9
    # zero padding reproduces |A theta|^2 with the reference's truncated matrix
    co = oracle.CostOracle(1, 8, np.zeros((1, 16), dtype=np.uint8), start=(0,), goal=(9,),
                           weights=(0, 0, 1), pad_mode=oracle.PAD_ZERO)
    theta = np.array([1.0, 2.5, 2.0, 4.0, 7.0, 6.5, 8.0, 8.5])
    A = oracle.differentiation_matrix(8, 2)
    c, terms = co.cost(theta)
    np.testing.assert_allclose(terms[2], np.sum((A @ theta) ** 2), rtol=1e-13)


def test_grid_generators(oracle):
    occ = oracle.two_bar_map(100)
    # the reference's cells: rows 0..59 x cols 65..74 and rows 40..98 x cols 35..44
    want = np.zeros((100, 100), dtype=np.uint8)
    want[0:60, 65:75] = 1
    want[40:99, 35:45] = 1
    np.testing.assert_array_equal(occ, want)
    assert occ[0, 99] == 0 and occ[99, 0] == 0   # start (x=99,y=0) and goal (x=0,y=99) are free
    sd, d = oracle.ssedt8(occ)
    ex = oracle.edt_exact(occ)
    assert np.all(d[occ == 1] == 0)
    # 8SSEDT is exact on most cells and never below the true distance
    assert np.all(d >= ex - 1e-4) and np.mean(np.abs(d - ex) < 1e-4) > 0.95
    assert sd[0, 70] < 0 and sd[0, 0] > 0
    # brute-force check of the exact EDT on a small random image
    rng = np.random.default_rng(0)
    small = (rng.random((12, 17)) < 0.1).astype(np.uint8)
    small[3, 4] = 1
    ys, xs = np.nonzero(small)
    yy, xx = np.mgrid[0:12, 0:17]
    brute = np.sqrt(((yy[..., None] - ys) ** 2 + (xx[..., None] - xs) ** 2).min(axis=-1))
    np.testing.assert_allclose(oracle.edt_exact(small), brute.astype(np.float32), rtol=1e-6)


def test_sepcma_restatement_converges_on_an_ellipsoid(oracle):
    """SepCMA (lmcma.cpp:466-577): its traces against the compiled reference are in
    tests/test_reference_pins.py; this checks that the restatement optimises."""
    n = 10
    scale = np.arange(1, n + 1, dtype=np.float64)
    s = oracle.OracleSepCMA(n, 12, initial=np.full(n, 3.0), sigma=1.0, seed=5)
    assert s.lam == 12 and np.all(s.diag == 1.0)
    for _ in range(300):
        s.tell_all(np.sum(scale * (s.arx - 1.0) ** 2, axis=1))
    assert np.abs(s.xmean - 1.0).max() < 1e-9 and s.sigma < 1e-6


def test_cmachol_restatement_converges_on_a_rotated_ellipsoid(oracle):
    """CMAChol (lmcma.cpp:579-754): pinned like SepCMA (tests/test_reference_pins.py); here the
    restatement must optimise a problem a diagonal model cannot (a rotated ellipsoid) and keep A A^-1 = I."""
    n = 8
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    H = Q @ np.diag(10.0 ** np.linspace(0, 3, n)) @ Q.T
    c = oracle.OracleCMAChol(n, 16, initial=np.full(n, 2.0), sigma=1.0, seed=3)
    for _ in range(600):
        X = c.arx - 1.0
        c.tell_all(np.einsum("ij,jk,ik->i", X, H, X))
    assert np.abs(c.xmean - 1.0).max() < 1e-6 and c.sigma < 1e-3
