"""Host-side logic of the product library against the oracle, and the shape of
the C ABI.  No GPU: nothing here launches a kernel or creates a context."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from lmcma_path_planner_amd import _lib
    return _lib.load()


def _declared_symbols():
    names = set()
    for header in ("lmcma_hip.h", "lmcma_opt.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(lmcma_(?:hip|opt|host|grid)_\w+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(lib):
    from lmcma_path_planner_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 50
    for name in sorted(declared):
        assert hasattr(lib, name), "liblmcma_hip.so does not export " + name
    # and the ctypes prototypes cover exactly the declared set
    assert set(_lib.SIGNATURES) == declared


def test_config_struct_matches_header_layout():
    from lmcma_path_planner_amd._lib import HipConfig
    # 14 int32 + uint64 + 3 doubles + 2*3 doubles + pointer
    assert ctypes.sizeof(HipConfig) == 14 * 4 + 8 + 3 * 8 + 6 * 8 + 8


def test_host_rng_is_bit_exact_with_oracle(oracle):
    from lmcma_path_planner_amd import planner
    for seed in (1, 7, 42, 2147483646):
        np.testing.assert_array_equal(planner.gauss_stream(seed, 2000), oracle.gauss_stream(seed, 2000))
        np.testing.assert_array_equal(planner.uniform_stream(seed, 500), oracle.uniform_stream(seed, 500))


def test_host_rng_matches_recorded_reference_values(golden):
    from lmcma_path_planner_amd import planner
    want = list(reversed(golden["rng_gauss_seed7_first3_as_printed"]))
    assert [float(v) for v in planner.gauss_stream(7, 3)] == want


def test_host_sort_matches_oracle(oracle):
    from lmcma_path_planner_amd import planner
    rng = np.random.default_rng(3)
    v = np.round(rng.normal(size=500), 1)  # many ties
    v[17] = np.nan
    s1, i1 = planner.stable_sort(v)
    s2, i2 = oracle.sort(v)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(s1[:-1], s2[:-1])


def test_smoothness_matrices_match_oracle(oracle):
    from lmcma_path_planner_amd import planner
    for order in range(4):
        np.testing.assert_array_equal(planner.differentiationMatrix(9, order), oracle.differentiation_matrix(9, order))
    for D, W in ((2, 1), (1, 8), (2, 32), (3, 12)):
        C1, C2 = planner.covariance(D, W), oracle.covariance(D, W)
        np.testing.assert_allclose(C1, C2, rtol=1e-10, atol=1e-14)
        L1, L2 = planner.cholesky(C2), oracle.cholesky(C2)
        np.testing.assert_allclose(L1, L2, rtol=1e-12, atol=1e-15)
        z = np.linspace(-1, 1, D * W)
        np.testing.assert_allclose(planner.applyCovL(L2, z), oracle.apply_cov_l(L2, z), rtol=1e-13, atol=1e-15)


def test_grid_generators_match_oracle(oracle):
    from lmcma_path_planner_amd import grids
    for size, nz in ((100, 1), (512, 1), (37, 1), (32, 32)):
        np.testing.assert_array_equal(grids.two_bar_map(size, nz), oracle.two_bar_map(size, nz))
    start, goal = [63.0, 0.0], [0.0, 63.0]
    base = oracle.two_bar_map(64)
    a = grids.add_random_boxes(base.copy(), 99, 0.2, start, goal)
    b = oracle.random_boxes(base, 99, 0.2, start, goal)
    np.testing.assert_array_equal(a, b)
    assert 0.19 < a.mean() < 0.3 and a[0, 63] == 0 and a[63, 0] == 0
    np.testing.assert_array_equal(grids.edt(a), oracle.edt_exact(a))
    sd1, d1 = grids.ssedt8(a)
    sd2, d2 = oracle.ssedt8(a)
    np.testing.assert_array_equal(sd1, sd2)
    np.testing.assert_array_equal(d1, d2)
    cost = grids.state_cost_from_distance(grids.edt(a), 0.5)
    assert cost.max() == 2.0 and cost.min() > 0
    # 3-D
    s3, g3 = [15.0, 0.0, 0.0], [0.0, 15.0, 15.0]
    base3 = oracle.two_bar_map(16, 16)
    a3 = grids.add_random_boxes(base3.copy(), 5, 0.2, s3, g3)
    np.testing.assert_array_equal(a3, oracle.random_boxes(base3, 5, 0.2, s3, g3))
    np.testing.assert_array_equal(grids.edt(a3), oracle.edt_exact(a3))


def test_benchmark_problem_shapes():
    from lmcma_path_planner_amd import grids
    p1 = grids.benchmark_problem(1)
    assert (p1.N, p1.lam, p1.m, p1.occ.shape) == (64, 16, 4, (100, 100))
    p2 = grids.benchmark_problem(2)
    assert (p2.N, p2.lam, p2.m, p2.occ.shape) == (128, 4096, 16, (512, 512))
    assert p2.algorithmic_bytes_per_candidate() == 1308.0
    assert p2.x0.shape == (128,) and p2.x0[0] < 511 and p2.x0[64] > 0
    assert p2.occ[0, 511] == 0 and p2.occ[511, 0] == 0


def test_missing_library_fails_loudly(monkeypatch):
    from lmcma_path_planner_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblmcma_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "lmcma_path_planner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "lmcma_oracle" not in text and "orc_" not in text, f
