"""ctypes binding of the CPU oracle (oracle/lmcma_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg as the checker / reported baseline.  Nothing under
lmcma_path_planner_amd/ imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblmcma_oracle.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_u8_p = ctypes.POINTER(ctypes.c_uint8)
c_u32_p = ctypes.POINTER(ctypes.c_uint32)

Z_REFERENCE, Z_PHILOX, Z_GIVEN = 0, 1, 2
GRID_BITS, GRID_F32 = 0, 1
PAD_ENDPOINT, PAD_ZERO = 0, 1


class Rng(ctypes.Structure):
    _fields_ = [("startseed", ctypes.c_long), ("aktseed", ctypes.c_long), ("aktrand", ctypes.c_long),
                ("rgrand", ctypes.c_long * 32), ("flgstored", ctypes.c_short), ("hold", ctypes.c_double)]


class ProblemC(ctypes.Structure):
    _fields_ = [("dims", ctypes.c_int), ("waypoints", ctypes.c_int), ("nx", ctypes.c_int), ("ny", ctypes.c_int),
                ("nz", ctypes.c_int), ("grid_type", ctypes.c_int), ("grid", ctypes.c_void_p),
                ("c_obs", ctypes.c_double), ("start", ctypes.c_double * 3), ("goal", ctypes.c_double * 3),
                ("w_len", ctypes.c_double), ("w_grid", ctypes.c_double), ("w_smooth", ctypes.c_double),
                ("pad_mode", ctypes.c_int)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    vp, i, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.orc_rng_start.argtypes = [ctypes.POINTER(Rng), ctypes.c_ulong]
    L.orc_rng_start.restype = ctypes.c_long
    L.orc_rng_uniform.argtypes = [ctypes.POINTER(Rng)]
    L.orc_rng_uniform.restype = d
    L.orc_rng_gauss.argtypes = [ctypes.POINTER(Rng)]
    L.orc_rng_gauss.restype = d
    L.orc_philox4x32_10.argtypes = [c_u32_p, c_u32_p, c_u32_p]
    L.orc_philox_normals.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, i, c_double_p]
    L.orc_sort.argtypes = [i, c_double_p, c_int_p]
    L.orc_differentiation_matrix.argtypes = [i, i, d, c_double_p, i]
    L.orc_covariance.argtypes = [i, i, c_double_p]
    L.orc_invert.argtypes = [c_double_p, c_double_p, i]
    L.orc_cholesky.argtypes = [c_double_p, c_double_p, i]
    L.orc_apply_cov_l.argtypes = [c_double_p, c_double_p, i]
    L.orc_lmcma_create.restype = vp
    L.orc_lmcma_create.argtypes = [i, i, i, c_double_p, c_double_p, c_double_p, d, c_double_p, ctypes.c_long, i]
    L.orc_lmcma_destroy.argtypes = [vp]
    L.orc_lmcma_ask.argtypes = [vp, c_double_p]
    L.orc_lmcma_tell.argtypes = [vp, c_double_p, i]
    L.orc_lmcma_tell_all.argtypes = [vp, c_double_p]
    L.orc_lmcma_update_only.argtypes = [vp, c_double_p]
    L.orc_lmcma_update_with_mean.argtypes = [vp, c_double_p, c_double_p]
    L.orc_lmcma_sample_given.argtypes = [vp, c_double_p]
    L.orc_lmcma_sample.argtypes = [vp]
    L.orc_lmcma_set_arx.argtypes = [vp, c_double_p]
    L.orc_sep_create.argtypes = [i, i, c_double_p, c_double_p, c_double_p, d, c_double_p, ctypes.c_long]
    L.orc_sep_create.restype = vp
    L.orc_sep_destroy.argtypes = [vp]
    L.orc_sep_tell_all.argtypes = [vp, c_double_p]
    L.orc_sep_set_arx.argtypes = [vp, c_double_p]
    L.orc_sep_lambda.argtypes, L.orc_sep_lambda.restype = [vp], i
    L.orc_sep_sigma.argtypes, L.orc_sep_sigma.restype = [vp], d
    for name in ("arx", "xmean", "diag"):
        f = getattr(L, "orc_sep_" + name)
        f.argtypes, f.restype = [vp], c_double_p
    L.orc_chol_create.argtypes = [i, i, c_double_p, c_double_p, c_double_p, d, c_double_p, ctypes.c_long]
    L.orc_chol_create.restype = vp
    L.orc_chol_destroy.argtypes = [vp]
    L.orc_chol_tell_all.argtypes = [vp, c_double_p]
    L.orc_chol_set_arx.argtypes = [vp, c_double_p]
    L.orc_chol_lambda.argtypes, L.orc_chol_lambda.restype = [vp], i
    L.orc_chol_sigma.argtypes, L.orc_chol_sigma.restype = [vp], d
    for name in ("arx", "xmean", "factor"):
        f = getattr(L, "orc_chol_" + name)
        f.argtypes, f.restype = [vp], c_double_p
    L.orc_lmcma_done.argtypes = [vp]
    L.orc_lmcma_done.restype = i
    for name in ("lambda", "mu", "m", "itr", "iterator_sz", "counteval"):
        f = getattr(L, "orc_lmcma_" + name)
        f.argtypes, f.restype = [vp], i
    for name in ("bestf", "sigma", "s", "mueff", "c1", "cc"):
        f = getattr(L, "orc_lmcma_" + name)
        f.argtypes, f.restype = [vp], d
    for name in ("arx", "xmean", "pc", "weights", "v_arr", "pc_arr", "nj", "lj", "sorted_fitness"):
        f = getattr(L, "orc_lmcma_" + name)
        f.argtypes, f.restype = [vp], c_double_p
    for name in ("iterator", "arindex"):
        f = getattr(L, "orc_lmcma_" + name)
        f.argtypes, f.restype = [vp], c_int_p
    L.orc_cost.argtypes = [ctypes.POINTER(ProblemC), c_double_p, c_double_p]
    L.orc_cost.restype = d
    L.orc_cost_callback.argtypes = [vp, c_double_p, i]
    L.orc_cost_callback.restype = d
    L.orc_cost_rows_f32.argtypes = [ctypes.POINTER(ProblemC), c_float_p, i, c_double_p, c_double_p]
    L.orc_lmcma_run_trajectory.argtypes = [vp, ctypes.POINTER(ProblemC), i]
    L.orc_set_threads.argtypes = [i]
    L.orc_get_threads.restype = i
    L.orc_toy_cost.argtypes = [d, d]
    L.orc_toy_cost.restype = d
    L.orc_two_bar_map.argtypes = [i, i, c_u8_p]
    L.orc_random_boxes.argtypes = [i, i, i, ctypes.c_uint64, d, c_double_p, c_double_p, c_u8_p]
    L.orc_pack_bits.argtypes = [c_u8_p, ctypes.c_long, c_u32_p]
    L.orc_8ssedt.argtypes = [c_u8_p, i, i, c_int_p, c_double_p]
    L.orc_edt_exact.argtypes = [c_u8_p, i, i, i, c_float_p]
    _lib = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def gauss_stream(seed, n):
    L = load()
    r = Rng()
    L.orc_rng_start(ctypes.byref(r), seed)
    return np.array([L.orc_rng_gauss(ctypes.byref(r)) for _ in range(n)])


def uniform_stream(seed, n):
    L = load()
    r = Rng()
    L.orc_rng_start(ctypes.byref(r), seed)
    return np.array([L.orc_rng_uniform(ctypes.byref(r)) for _ in range(n)])


def philox(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    load().orc_philox4x32_10(c, k, o)
    return list(o)


def philox_normals(seed, generation, candidate, N):
    z = np.empty(N)
    load().orc_philox_normals(seed, generation, candidate, N, _dp(z))
    return z


def sort(values):
    v = _f64(values).copy()
    idx = np.empty(v.size, dtype=np.int32)
    load().orc_sort(v.size, _dp(v), idx.ctypes.data_as(c_int_p))
    return v, idx


def differentiation_matrix(n, order, dt=1.0):
    out = np.zeros((n, n))
    load().orc_differentiation_matrix(n, order, dt, _dp(out), -1)
    return out


def covariance(dims, waypoints):
    n = dims * waypoints
    out = np.empty((n, n))
    rc = load().orc_covariance(dims, waypoints, _dp(out))
    assert rc == 0
    return out


def cholesky(C):
    C = _f64(C)
    n = C.shape[0]
    L = np.empty((n, n))
    assert load().orc_cholesky(_dp(C), _dp(L), n) == 0
    return L


def apply_cov_l(L, z):
    z = _f64(z).copy()
    load().orc_apply_cov_l(_dp(_f64(L)), _dp(z), z.size)
    return z


class OracleSepCMA:
    """The fp64 restatement of the reference's SepCMA (lmcma.cpp:466-577)."""

    def __init__(self, N, lam=0, initial=None, lo=None, hi=None, sigma=1.0, cov=None, seed=42):
        self.L = load()
        self.N = N
        init, lo, hi, cov = _f64(initial), _f64(lo), _f64(hi), _f64(cov)
        self.h = self.L.orc_sep_create(N, lam, _dp(init), _dp(lo), _dp(hi), sigma, _dp(cov), seed)
        self.lam = self.L.orc_sep_lambda(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_sep_destroy(self.h)
            self.h = None

    def tell_all(self, fitness):
        self.L.orc_sep_tell_all(self.h, _dp(_f64(fitness)))

    def set_arx(self, rows):
        rows = _f64(rows)
        assert rows.shape == (self.lam, self.N)
        self.L.orc_sep_set_arx(self.h, _dp(rows))

    def _arr(self, name, shape):
        return np.ctypeslib.as_array(getattr(self.L, "orc_sep_" + name)(self.h), shape=shape).copy()

    @property
    def arx(self):
        return self._arr("arx", (self.lam, self.N))

    @property
    def xmean(self):
        return self._arr("xmean", (self.N,))

    @property
    def diag(self):
        return self._arr("diag", (self.N,))

    @property
    def sigma(self):
        return self.L.orc_sep_sigma(self.h)


class OracleCMAChol:
    """The fp64 restatement of the reference's CMAChol (lmcma.cpp:579-754)."""

    def __init__(self, N, lam=0, initial=None, lo=None, hi=None, sigma=1.0, cov=None, seed=42):
        self.L = load()
        self.N = N
        init, lo, hi, cov = _f64(initial), _f64(lo), _f64(hi), _f64(cov)
        self.h = self.L.orc_chol_create(N, lam, _dp(init), _dp(lo), _dp(hi), sigma, _dp(cov), seed)
        self.lam = self.L.orc_chol_lambda(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_chol_destroy(self.h)
            self.h = None

    def tell_all(self, fitness):
        self.L.orc_chol_tell_all(self.h, _dp(_f64(fitness)))

    def set_arx(self, rows):
        rows = _f64(rows)
        assert rows.shape == (self.lam, self.N)
        self.L.orc_chol_set_arx(self.h, _dp(rows))

    def _arr(self, name, shape):
        return np.ctypeslib.as_array(getattr(self.L, "orc_chol_" + name)(self.h), shape=shape).copy()

    @property
    def arx(self):
        return self._arr("arx", (self.lam, self.N))

    @property
    def xmean(self):
        return self._arr("xmean", (self.N,))

    @property
    def factor(self):
        return self._arr("factor", (self.N, self.N))

    @property
    def sigma(self):
        return self.L.orc_chol_sigma(self.h)


class OracleLMCMA:
    """The fp64 restatement of the reference optimiser (m decoupled from lambda)."""

    def __init__(self, N, lam=0, m=0, initial=None, lo=None, hi=None, sigma=1.0, cov=None, seed=42,
                 z_source=Z_REFERENCE):
        self.L = load()
        self.N = N
        init, lo, hi, cov = _f64(initial), _f64(lo), _f64(hi), _f64(cov)
        self.h = self.L.orc_lmcma_create(N, lam, m, _dp(init), _dp(lo), _dp(hi), sigma, _dp(cov), seed, z_source)
        self.lam = self.L.orc_lmcma_lambda(self.h)
        self.mu = self.L.orc_lmcma_mu(self.h)
        self.m = self.L.orc_lmcma_m(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_lmcma_destroy(self.h)
            self.h = None

    def ask(self):
        out = np.empty(self.N)
        self.L.orc_lmcma_ask(self.h, _dp(out))
        return out

    def tell(self, feedbacks):
        f = _f64(np.atleast_1d(feedbacks))
        self.L.orc_lmcma_tell(self.h, _dp(f), f.size)

    def tell_all(self, fitness):
        self.L.orc_lmcma_tell_all(self.h, _dp(_f64(fitness)))

    def update_only(self, fitness):
        self.L.orc_lmcma_update_only(self.h, _dp(_f64(fitness)))

    def update_with_mean(self, fitness, xmean_new):
        self.L.orc_lmcma_update_with_mean(self.h, _dp(_f64(fitness)), _dp(_f64(xmean_new)))

    def sample_given(self, Z):
        self.L.orc_lmcma_sample_given(self.h, _dp(_f64(Z)))

    def sample(self):
        self.L.orc_lmcma_sample(self.h)

    def set_arx(self, rows):
        rows = _f64(rows)
        assert rows.shape == (self.lam, self.N)
        self.L.orc_lmcma_set_arx(self.h, _dp(rows))

    def run_trajectory(self, cost_oracle, generations):
        """generations of ask / score (fp64 cost of the float-rounded rows) / tell, all in C."""
        self.L.orc_lmcma_run_trajectory(self.h, ctypes.byref(cost_oracle.p), int(generations))

    def _arr(self, name, shape, dtype=np.float64):
        p = getattr(self.L, "orc_lmcma_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    @property
    def arx(self):
        return self._arr("arx", (self.lam, self.N))

    @property
    def xmean(self):
        return self._arr("xmean", (self.N,))

    @property
    def pc(self):
        return self._arr("pc", (self.N,))

    @property
    def weights(self):
        return self._arr("weights", (self.mu,))

    @property
    def v_arr(self):
        return self._arr("v_arr", (self.m, self.N))

    @property
    def pc_arr(self):
        return self._arr("pc_arr", (self.m, self.N))

    @property
    def nj(self):
        return self._arr("nj", (self.m,))

    @property
    def lj(self):
        return self._arr("lj", (self.m,))

    @property
    def iterator(self):
        return self._arr("iterator", (self.m,))

    @property
    def arindex(self):
        return self._arr("arindex", (self.lam,))

    @property
    def sorted_fitness(self):
        return self._arr("sorted_fitness", (self.lam,))

    def done(self):
        """LMCMA::isBehaviorLearningDone (lmcma.cpp:426-429): sigma < 1e-20."""
        return bool(self.L.orc_lmcma_done(self.h))

    def scalar(self, name):
        return getattr(self.L, "orc_lmcma_" + name)(self.h)

    sigma = property(lambda self: self.scalar("sigma"))
    s = property(lambda self: self.scalar("s"))
    itr = property(lambda self: self.scalar("itr"))
    iterator_sz = property(lambda self: self.scalar("iterator_sz"))
    counteval = property(lambda self: self.scalar("counteval"))
    BestF = property(lambda self: self.scalar("bestf"))
    mueff = property(lambda self: self.scalar("mueff"))
    c1 = property(lambda self: self.scalar("c1"))
    cc = property(lambda self: self.scalar("cc"))


class CostOracle:
    """fp64 trajectory cost (the cost contract) over a bit-packed or float grid."""

    def __init__(self, dims, waypoints, cells, start, goal, weights=(1.0, 1000.0, 1.0), c_obs=1.0,
                 pad_mode=PAD_ENDPOINT):
        self.L = load()
        cells = np.ascontiguousarray(cells)
        shape = cells.shape
        nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
        p = ProblemC()
        p.dims, p.waypoints, p.nx, p.ny, p.nz = dims, waypoints, nx, ny, nz
        if cells.dtype == np.uint8:
            words = np.zeros((cells.size + 31) // 32, dtype=np.uint32)
            self.L.orc_pack_bits(cells.ctypes.data_as(c_u8_p), cells.size, words.ctypes.data_as(c_u32_p))
            self._keep = words
            p.grid_type = GRID_BITS
        elif cells.dtype == np.float32:
            self._keep = cells
            p.grid_type = GRID_F32
        else:
            raise TypeError("cells must be uint8 occupancy or float32 state cost")
        p.grid = self._keep.ctypes.data_as(ctypes.c_void_p)
        p.c_obs = c_obs
        for d in range(3):
            p.start[d] = float(start[d]) if d < len(start) else 0.0
            p.goal[d] = float(goal[d]) if d < len(goal) else 0.0
        p.w_len, p.w_grid, p.w_smooth = weights
        p.pad_mode = pad_mode
        self.p = p
        self.N = dims * waypoints

    def cost(self, row):
        row = _f64(row)
        terms = np.empty(3)
        c = self.L.orc_cost(ctypes.byref(self.p), _dp(row), _dp(terms))
        return c, terms

    def cost_rows_f32(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n = rows.shape[0]
        cost = np.empty(n)
        terms = np.empty((n, 3))
        self.L.orc_cost_rows_f32(ctypes.byref(self.p), rows.ctypes.data_as(c_float_p), n, _dp(cost), _dp(terms))
        return cost, terms


def set_threads(n):
    """Host threads for the oracle's loops over candidates (results do not depend on it)."""
    load().orc_set_threads(int(n))


def toy_cost(x, y):
    return load().orc_toy_cost(float(x), float(y))


def two_bar_map(size, nz=1):
    occ = np.zeros((nz, size, size), dtype=np.uint8)
    load().orc_two_bar_map(size, nz, occ.ctypes.data_as(c_u8_p))
    return occ if nz > 1 else occ[0]


def random_boxes(occ, seed, fill, start, goal):
    occ = np.ascontiguousarray(occ, dtype=np.uint8).copy()
    shape = occ.shape
    nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
    s = np.asarray(list(start) + [0.0] * (3 - len(start)), dtype=np.float64)
    g = np.asarray(list(goal) + [0.0] * (3 - len(goal)), dtype=np.float64)
    load().orc_random_boxes(nx, ny, nz, seed, fill, _dp(s), _dp(g), occ.ctypes.data_as(c_u8_p))
    return occ


def ssedt8(inside):
    inside = np.ascontiguousarray(inside, dtype=np.uint8)
    h, w = inside.shape
    sd = np.empty((h, w), dtype=np.int32)
    d = np.empty((h, w))
    load().orc_8ssedt(inside.ctypes.data_as(c_u8_p), w, h, sd.ctypes.data_as(c_int_p), _dp(d))
    return sd, d


def edt_exact(occ):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    shape = occ.shape
    nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
    out = np.empty(shape, dtype=np.float32)
    load().orc_edt_exact(occ.ctypes.data_as(c_u8_p), nx, ny, nz, out.ctypes.data_as(c_float_p))
    return out
