"""ctypes binding of oracle/_ref/liblmcma_ref.so: the reference's own lmcma.cpp, compiled where it
lies by `make -C oracle ref` (see oracle/Makefile and oracle/ref_harness.cpp for what is and is not
part of that build).

TEST INFRASTRUCTURE ONLY: used by tests/golden/make_reference_fixtures.py to write the fixtures that
pin the oracle, by the CPU tests for live comparisons when the library exists, and by bench.py's
cpu_baseline leg.  Nothing under lmcma_path_planner_amd/ imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "liblmcma_ref.so")
REFERENCE_SOURCE = "/root/reference/lmcma_path_planner/src/lmcma.cpp"

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
COST_FN = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, c_double_p, ctypes.c_int)

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def build():
    """Compile the reference (only possible where the reference tree exists)."""
    if not os.path.exists(REFERENCE_SOURCE):
        raise FileNotFoundError(REFERENCE_SOURCE)
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    L = ctypes.CDLL(LIB_PATH)
    vp, i, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.ref_uniform_stream.argtypes = [ctypes.c_long, i, c_double_p]
    L.ref_gauss_stream.argtypes = [ctypes.c_long, i, c_double_p]
    L.ref_sort.argtypes = [i, c_double_p, c_int_p]
    L.ref_differentiation_matrix.argtypes = [i, i, d, c_double_p, i]
    for kind in ("lmcma", "sep", "chol"):
        f = getattr(L, "ref_%s_create" % kind)
        f.argtypes, f.restype = [i, i, c_double_p, c_double_p, c_double_p, d, i], vp
        getattr(L, "ref_%s_destroy" % kind).argtypes = [vp]
        getattr(L, "ref_%s_tell_all" % kind).argtypes = [vp, c_double_p]
    L.ref_lmcma_ask.argtypes = [vp, c_double_p]
    L.ref_lmcma_tell.argtypes = [vp, c_double_p, i]
    L.ref_lmcma_done.argtypes, L.ref_lmcma_done.restype = [vp], i
    for name in ("lambda", "mu", "nvectors", "itr", "iterator_sz", "counteval"):
        f = getattr(L, "ref_lmcma_" + name)
        f.argtypes, f.restype = [vp], i
    for name in ("bestf", "sigma", "s", "mueff", "c1", "cc", "K", "M"):
        f = getattr(L, "ref_lmcma_" + name)
        f.argtypes, f.restype = [vp], d
    for name in ("arx", "arfitness", "prev_arfitness", "xmean", "pc", "weights", "v_arr", "pc_arr", "nj", "lj"):
        f = getattr(L, "ref_lmcma_" + name)
        f.argtypes, f.restype = [vp], c_double_p
    for name in ("iterator", "arindex", "t", "vec"):
        f = getattr(L, "ref_lmcma_" + name)
        f.argtypes, f.restype = [vp], c_int_p
    for kind, arrays in (("sep", ("arx", "xmean", "diag")), ("chol", ("arx", "xmean", "factor"))):
        f = getattr(L, "ref_%s_lambda" % kind)
        f.argtypes, f.restype = [vp], i
        f = getattr(L, "ref_%s_sigma" % kind)
        f.argtypes, f.restype = [vp], d
        for name in arrays:
            f = getattr(L, "ref_%s_%s" % (kind, name))
            f.argtypes, f.restype = [vp], c_double_p
    L.ref_lmcma_run_timed.argtypes = [vp, ctypes.c_void_p, ctypes.c_void_p, i]
    L.ref_lmcma_run_timed.restype = d
    _lib = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def uniform_stream(seed, n):
    out = np.empty(n)
    load().ref_uniform_stream(seed, n, _dp(out))
    return out


def gauss_stream(seed, n):
    out = np.empty(n)
    load().ref_gauss_stream(seed, n, _dp(out))
    return out


def sort(values):
    v = _f64(values).copy()
    idx = np.empty(v.size, dtype=np.int32)
    load().ref_sort(v.size, _dp(v), idx.ctypes.data_as(c_int_p))
    return v, idx


def differentiation_matrix(n, order, dt=1.0):
    out = np.zeros((n, n))
    load().ref_differentiation_matrix(n, order, dt, _dp(out), -1)
    return out


class _Base:
    kind = None
    arrays = {}

    def __init__(self, N, lam=0, initial=None, lo=None, hi=None, sigma=1.0, seed=42):
        self.L = load()
        self.N = N
        init, lo, hi = _f64(initial), _f64(lo), _f64(hi)
        self.h = getattr(self.L, "ref_%s_create" % self.kind)(N, lam, _dp(init), _dp(lo), _dp(hi), sigma, seed)
        self.lam = getattr(self.L, "ref_%s_lambda" % self.kind)(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, "ref_%s_destroy" % self.kind)(self.h)
            self.h = None

    def tell_all(self, fitness):
        f = _f64(fitness)
        assert f.size == self.lam
        getattr(self.L, "ref_%s_tell_all" % self.kind)(self.h, _dp(f))

    def _arr(self, name, shape):
        p = getattr(self.L, "ref_%s_%s" % (self.kind, name))(self.h)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    def scalar(self, name):
        return getattr(self.L, "ref_%s_%s" % (self.kind, name))(self.h)

    sigma = property(lambda self: self.scalar("sigma"))
    arx = property(lambda self: self._arr("arx", (self.lam, self.N)))
    xmean = property(lambda self: self._arr("xmean", (self.N,)))


class RefLMCMA(_Base):
    """The reference's class LMCMA (lmcma.hpp:89-144); the number of direction vectors is lambda."""
    kind = "lmcma"

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.mu = self.scalar("mu")
        self.m = self.scalar("nvectors")

    def ask(self):
        out = np.empty(self.N)
        self.L.ref_lmcma_ask(self.h, _dp(out))
        return out

    def tell(self, feedbacks):
        f = _f64(np.atleast_1d(feedbacks)).copy()
        self.L.ref_lmcma_tell(self.h, _dp(f), f.size)

    def done(self):
        return bool(self.L.ref_lmcma_done(self.h))

    def run_timed(self, cost_fn_ptr, user_ptr, generations):
        """seconds for `generations` generations of the reference's per-candidate loop; cost_fn_ptr is
        the address of a C function double(void* user, const double* row, int N)."""
        return self.L.ref_lmcma_run_timed(self.h, cost_fn_ptr, user_ptr, int(generations))

    pc = property(lambda self: self._arr("pc", (self.N,)))
    weights = property(lambda self: self._arr("weights", (self.mu,)))
    v_arr = property(lambda self: self._arr("v_arr", (self.m, self.N)))
    pc_arr = property(lambda self: self._arr("pc_arr", (self.m, self.N)))
    nj = property(lambda self: self._arr("nj", (self.m,)))
    lj = property(lambda self: self._arr("lj", (self.m,)))
    iterator = property(lambda self: self._arr("iterator", (self.m,)))
    arindex = property(lambda self: self._arr("arindex", (self.lam,)))
    sorted_fitness = property(lambda self: self._arr("prev_arfitness", (self.lam,)))
    s = property(lambda self: self.scalar("s"))
    itr = property(lambda self: self.scalar("itr"))
    iterator_sz = property(lambda self: self.scalar("iterator_sz"))
    counteval = property(lambda self: self.scalar("counteval"))
    BestF = property(lambda self: self.scalar("bestf"))
    mueff = property(lambda self: self.scalar("mueff"))
    c1 = property(lambda self: self.scalar("c1"))
    cc = property(lambda self: self.scalar("cc"))
    K = property(lambda self: self.scalar("K"))
    M = property(lambda self: self.scalar("M"))


class RefSepCMA(_Base):
    kind = "sep"
    diag = property(lambda self: self._arr("diag", (self.N,)))


class RefCMAChol(_Base):
    kind = "chol"
    factor = property(lambda self: self._arr("factor", (self.N, self.N)))
