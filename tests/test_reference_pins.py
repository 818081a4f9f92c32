"""Pins of the CPU oracle: oracle/lmcma_oracle.c replayed on the inputs of tests/golden/reference_kat.npz
must give the bits the REFERENCE's own lmcma.cpp gave (the fixture was written by
tests/golden/make_reference_fixtures.py from the reference compiled where it lies; the reference itself
does not travel).  CPU only.  Where oracle/_ref/liblmcma_ref.so exists (the build container) the same
comparison also runs live on further seeds.

Not pinned by any of this: covariance()/invert/cholesky/applyCovL (Eigen, absent), the trajectory cost
(its integration lives in OMPL, not vendored) and every run with fewer direction vectors than lambda
(the reference has no such mode, lmcma.cpp:266)."""
import json
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float64).tobytes())


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_kat.npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def test_random_streams(oracle, kat):
    z, _ = kat
    for seed in (1, 7, 42, 2147483646):
        assert same_bits(oracle.uniform_stream(seed, 96), z["rng/uniform/%d" % seed])
        assert same_bits(oracle.gauss_stream(seed, 97), z["rng/gauss/%d" % seed])


def test_constants(oracle, kat):
    z, meta = kat
    for c in meta["constants"]:
        N, lam = c["N"], c["lam_arg"]
        o = oracle.OracleLMCMA(N, lam, 0, initial=np.zeros(N), seed=1)
        assert (o.lam, o.mu, o.m) == (c["lam"], c["mu"], c["nvectors"])
        assert same_bits(o.weights, z["const/%d_%d/weights" % (N, lam)])
        mueff, c1, cc, K, M = z["const/%d_%d/scalars" % (N, lam)]
        assert (o.mueff, o.c1, o.cc) == (mueff, c1, cc)
        assert 1.0 / np.sqrt(1.0 - o.c1) == K and np.sqrt(1.0 - o.c1) == M


def test_sort(oracle, kat):
    z, meta = kat
    for name in meta["sorts"]:
        sv, idx = oracle.sort(z["sort/%s/in" % name])
        assert same_bits(sv, z["sort/%s/sorted" % name])
        assert np.array_equal(idx, z["sort/%s/index" % name])


def test_differentiation_matrix(oracle, kat):
    z, meta = kat
    for n, order, dt in meta["diffmats"]:
        assert same_bits(oracle.differentiation_matrix(n, order, dt), z["diffmat/%d_%d_%g" % (n, order, dt)])


def _vec(v, N):
    return None if v is None else np.full(N, float(v))


def test_lmcma_traces(oracle, kat):
    z, meta = kat
    assert {(t["N"], t["lam"]) for t in meta["traces"]} >= {(2, 6), (8, 6), (64, 16), (128, 64)}
    for t in meta["traces"]:
        p = "trace/%s/" % t["name"]
        N = t["N"]
        o = oracle.OracleLMCMA(N, t["lam"], 0, initial=_vec(t["initial"], N), lo=_vec(t["lo"], N), hi=_vec(t["hi"], N),
                               sigma=t["sigma"], seed=t["seed"])
        assert o.m == o.lam == t["lam"]
        fed, xmean, sigma, s = z[p + "fed"], z[p + "xmean"], z[p + "sigma"], z[p + "s"]
        digests, iterators, iterator_sz = z[p + "arx_crc32"], z[p + "iterator"], z[p + "iterator_sz"]
        assert same_bits(o.xmean, xmean[0]) and crc(o.arx) == digests[0], t["name"]
        for g in range(t["generations"]):
            if g in t["arx_generations"]:
                assert same_bits(o.arx, z[p + "arx_g%d" % g]), (t["name"], g)
            o.tell_all(fed[g])
            assert same_bits(o.xmean, xmean[g + 1]), (t["name"], g)
            assert o.sigma == sigma[g + 1] and o.s == s[g + 1], (t["name"], g)
            assert o.BestF == z[p + "bestf"][g]
            assert crc(o.arx) == digests[g + 1], (t["name"], g)
            k = o.iterator_sz
            assert k == iterator_sz[g] and np.array_equal(o.iterator[:k], iterators[g, :k]), (t["name"], g)
        it = o.iterator[: o.iterator_sz]
        assert same_bits(o.v_arr[it], z[p + "final_v"]) and same_bits(o.pc_arr[it], z[p + "final_pc_arr"])
        assert same_bits(o.nj[it], z[p + "final_nj"]) and same_bits(o.lj[it], z[p + "final_lj"])
        assert same_bits(o.pc, z[p + "final_pc"])
        assert o.counteval == t["generations"] * t["lam"] and o.itr == t["generations"]
    # the store of direction vectors was recycled in the long traces (generations > lambda)
    assert any(t["generations"] > t["lam"] + 8 for t in meta["traces"])


def test_feedback_terms_are_summed_left_to_right(oracle, kat):
    z, _ = kat
    o = oracle.OracleLMCMA(4, 6, 0, initial=np.full(4, 1.0), sigma=0.7, seed=11)
    fed = z["terms/fed"]
    for g in range(fed.shape[0]):
        for i in range(6):
            o.ask()
            o.tell(fed[g, i])
        assert same_bits(o.xmean, z["terms/xmean"][g])
    assert o.BestF == z["terms/bestf"][0]


def test_stop_rule_fires_in_the_reference_generation(oracle, kat):
    z, meta = kat
    st = meta["stop"]
    o = oracle.OracleLMCMA(st["N"], st["lam"], 0, initial=np.full(st["N"], st["initial"]), sigma=st["sigma"], seed=st["seed"])
    fed = z["stop/fed"]
    assert fed.shape[0] == st["generation_done"] + 1
    for g in range(fed.shape[0]):
        assert not o.done()
        o.tell_all(fed[g])
    assert o.done() and o.sigma == z["stop/sigma_final"][0] < 1e-20


def test_sibling_optimisers(oracle, kat):
    z, meta = kat
    for t in meta["siblings"]:
        cls = oracle.OracleSepCMA if t["kind"] == "sep" else oracle.OracleCMAChol
        N = t["N"]
        o = cls(N, t["lam"], initial=_vec(t["initial"], N), lo=_vec(t["lo"], N), hi=_vec(t["hi"], N), sigma=t["sigma"],
                seed=t["seed"])
        p = "sibling/%s/" % t["name"]
        assert o.lam == t["lam"] and crc(o.arx) == z[p + "arx_crc32"][0]
        for g in range(t["generations"]):
            o.tell_all(z[p + "fed"][g])
            assert same_bits(o.xmean, z[p + "xmean"][g + 1]), (t["name"], g)
            assert o.sigma == z[p + "sigma"][g + 1], (t["name"], g)
            assert crc(o.arx) == z[p + "arx_crc32"][g + 1], (t["name"], g)
        assert same_bits(o.arx, z[p + "final_arx"])
        assert same_bits(o.diag if t["kind"] == "sep" else o.factor, z[p + "final_shape"])


def test_fixture_holds_the_earlier_probe(kat, golden):
    """The numbers recorded by hand before the reference could be built here are those of the fixture."""
    z, _ = kat
    assert list(z["rng/gauss/7"][:3]) == list(reversed(golden["rng_gauss_seed7_first3_as_printed"]))
    assert z["trace/sphere_8_6_probe/bestf"][-1] == golden["sphere_trace"]["BestF_after_300"]
    assert z["trace/sphere_8_6_probe/arx_g0"][0, 0] == golden["sphere_trace"]["first_candidate_first2"][0]


# ---- live, where the compiled reference exists (build container only) -----------------------------
def _pyref():
    from oracle import pyref
    if not pyref.available():
        pytest.skip("oracle/_ref/liblmcma_ref.so not built (the reference tree is absent on this machine)")
    return pyref


def test_live_reference_against_oracle_on_fresh_seeds(oracle):
    pyref = _pyref()
    rng = np.random.default_rng(2024)
    for case in range(12):
        N = int(rng.choice([2, 3, 8, 17, 64]))
        lam = int(rng.choice([0, 4, 6, 9, 16]))
        seed = int(rng.integers(1, 2147483646))
        bounded = bool(case % 2)
        lo, hi = (np.full(N, -1.0), np.full(N, 2.5)) if bounded else (None, None)
        init = rng.normal(size=N)
        w = 10.0 ** (2.0 * np.arange(N) / max(N - 1, 1))
        r = pyref.RefLMCMA(N, lam, initial=init, lo=lo, hi=hi, sigma=0.8, seed=seed)
        o = oracle.OracleLMCMA(N, lam, 0, initial=init, lo=lo, hi=hi, sigma=0.8, seed=seed)
        assert r.lam == o.lam
        for g in range(r.lam + 12):
            X = r.arx
            assert same_bits(X, o.arx), (case, g)
            f = (w * X * X).sum(axis=1)
            r.tell_all(f)
            o.tell_all(f)
            assert same_bits(r.xmean, o.xmean) and r.sigma == o.sigma and r.s == o.s
        k = r.iterator_sz
        assert k == o.iterator_sz and np.array_equal(r.iterator[:k], o.iterator[:k])
        it = r.iterator[:k]
        assert same_bits(r.v_arr[it], o.v_arr[it]) and same_bits(r.nj[it], o.nj[it]) and same_bits(r.lj[it], o.lj[it])
        assert r.BestF == o.BestF and r.counteval == o.counteval


def test_live_reference_siblings_against_oracle(oracle):
    pyref = _pyref()
    for rc, oc, shape in ((pyref.RefSepCMA, oracle.OracleSepCMA, "diag"), (pyref.RefCMAChol, oracle.OracleCMAChol, "factor")):
        for seed in (2, 77):
            r = rc(7, 10, initial=np.full(7, 1.5), sigma=0.6, seed=seed)
            o = oc(7, 10, initial=np.full(7, 1.5), sigma=0.6, seed=seed)
            for g in range(30):
                X = r.arx
                assert same_bits(X, o.arx)
                f = (X * X).sum(axis=1) + X[:, 0] * X[:, 1]
                r.tell_all(f)
                o.tell_all(f)
            assert same_bits(r.xmean, o.xmean) and r.sigma == o.sigma
            assert same_bits(getattr(r, shape), getattr(o, shape))
