"""Python mirror of the reference's optimiser interface.

``LMCMA`` has the constructor arguments and the method names of the reference's
``class LMCMA`` (lmcma_path_planner/src/lmcma.hpp:131-138, 72-73) and drives the
C++ class of include/lmcma.hpp through the C bindings of include/lmcma_opt.h, so
a loop written like the reference's example (lmcma_path_planner/src/example_lmcma.cpp:47-60)
reads the same here::

    opt = LMCMA(params, -1, lo, hi)
    opt.init(2)
    for _ in range(1000):
        x = opt.getNextParameterVector()
        opt.setEvaluationFeedback([cost(x)])
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import OptInfo, c_double_p, check_opt


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class LMCMA:
    _create = "lmcma_opt_create"

    def __init__(self, initialParams, lambda_=0, loBounds=None, hiBounds=None, sigma=1.0, covariance=None,
                 inseed=0, verbose=False):
        self.lib = _lib.load()
        init, lo, hi, cov = _f64(initialParams), _f64(loBounds), _f64(hiBounds), _f64(covariance)
        n = None
        for a in (init, lo, hi):
            if a is not None:
                n = a.size
        if n is None:
            raise ValueError("give initialParams or bounds so the dimension is known")
        self._n_hint = n
        self._h = ctypes.c_void_p()
        check_opt(getattr(self.lib, self._create)(_dp(init), _dp(lo), _dp(hi), _dp(cov), n, int(lambda_), float(sigma),
                                                  int(inseed), 1 if verbose else 0, ctypes.byref(self._h)),
                  self._create)
        self.N = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.lmcma_opt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- extensions of include/lmcma.hpp (before init) ---------------------------------
    def setOptions(self, m=0, device=0, philox=False, device_update=False):
        check_opt(self.lib.lmcma_opt_set_options(self._h, int(m), int(device), int(philox), int(device_update)),
                  "lmcma_opt_set_options")

    def setPopulationShard(self, world, rank, unique_id):
        """This object is shard `rank` of a population sharded over `world` processes (one GPU each)."""
        if len(unique_id) != 128:
            raise ValueError("the unique id has 128 bytes")
        check_opt(self.lib.lmcma_opt_set_shard(self._h, int(world), int(rank), bytes(unique_id)),
                  "lmcma_opt_set_shard")

    def setTrajectoryProblem(self, dims, waypoints, cells, start, goal, cell_type=None, c_obs=1.0, w_len=1.0,
                             w_grid=1000.0, w_smooth=1.0, pad_mode=0):
        cells = np.ascontiguousarray(cells)
        if cell_type is None:
            cell_type = {np.dtype(np.uint8): 1, np.dtype(np.float32): 2}[cells.dtype]
        shape = cells.shape
        nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
        s, g = _f64(list(start) + [0.0] * (3 - len(start))), _f64(list(goal) + [0.0] * (3 - len(goal)))
        check_opt(self.lib.lmcma_opt_set_problem(self._h, dims, waypoints, nx, ny, nz, cell_type,
                                                 cells.ctypes.data_as(ctypes.c_void_p), float(c_obs), _dp(s),
                                                 _dp(g), w_len, w_grid, w_smooth, pad_mode),
                  "lmcma_opt_set_problem")

    # ---- the reference's interface -------------------------------------------------------
    def init(self, N):
        self.N = int(N)
        check_opt(self.lib.lmcma_opt_init(self._h, self.N), "lmcma_opt_init")

    def getNextParameterVector(self):
        out = np.empty(self.N, dtype=np.float64)
        check_opt(self.lib.lmcma_opt_ask(self._h, _dp(out), self.N), "lmcma_opt_ask")
        return out

    def setEvaluationFeedback(self, feedbacks):
        f = _f64(np.atleast_1d(feedbacks))
        check_opt(self.lib.lmcma_opt_tell(self._h, _dp(f), f.size), "lmcma_opt_tell")

    def isBehaviorLearningDone(self):
        rc = self.lib.lmcma_opt_done(self._h)
        check_opt(rc, "lmcma_opt_done")
        return rc == 1

    def step(self, generations=1):
        check_opt(self.lib.lmcma_opt_step(self._h, int(generations)), "lmcma_opt_step")

    # ---- readers --------------------------------------------------------------------------
    def _info(self):
        info = OptInfo()
        check_opt(self.lib.lmcma_opt_get_info(self._h, ctypes.byref(info)), "lmcma_opt_get_info")
        return info

    @property
    def counteval(self):
        return self._info().counteval

    @property
    def BestF(self):
        return self._info().best_f

    @property
    def sigma(self):
        return self._info().sigma

    @property
    def generation(self):
        return self._info().generation

    @property
    def populationSize(self):
        return self._info().lambda_

    @property
    def directionVectors(self):
        return self._info().m

    def mean(self):
        out = np.empty(self.N, dtype=np.float64)
        check_opt(self.lib.lmcma_opt_get_mean(self._h, _dp(out)), "lmcma_opt_get_mean")
        return out

    def population(self):
        out = np.empty((self.populationSize, self.N), dtype=np.float64)
        check_opt(self.lib.lmcma_opt_get_population(self._h, _dp(out)), "lmcma_opt_get_population")
        return out

    def lastCosts(self):
        out = np.empty(self.populationSize, dtype=np.float64)
        check_opt(self.lib.lmcma_opt_get_last_costs(self._h, _dp(out)), "lmcma_opt_get_last_costs")
        return out


class SepCMA(LMCMA):
    """Separable CMA-ES with the reference's interface (lmcma.hpp:152-190); sampling and, with a
    trajectory problem, scoring on the device, the update on the host."""
    _create = "lmcma_opt_create_sep"

    def setOptions(self, device=0):
        check_opt(self.lib.lmcma_opt_set_options(self._h, 0, int(device), 0, 0), "lmcma_opt_set_options")

    def diagonal(self):
        out = np.empty(self.N, dtype=np.float64)
        check_opt(self.lib.lmcma_opt_get_diagonal(self._h, _dp(out)), "lmcma_opt_get_diagonal")
        return out


class CMAChol(LMCMA):
    """Cholesky-CMA-ES with the reference's interface (lmcma.hpp:199-239); the dense product A z and,
    with a trajectory problem, scoring on the device, the factor update on the host."""
    _create = "lmcma_opt_create_chol"

    def setOptions(self, device=0):
        check_opt(self.lib.lmcma_opt_set_options(self._h, 0, int(device), 0, 0), "lmcma_opt_set_options")

    def factor(self):
        out = np.empty((self.N, self.N), dtype=np.float64)
        check_opt(self.lib.lmcma_opt_get_factor(self._h, _dp(out)), "lmcma_opt_get_factor")
        return out


# ---- host helpers (reference free functions) ----------------------------------------------

def gauss_stream(seed, n):
    """n draws of random_Gauss after random_init(seed) (lmcma.cpp:35-82)."""
    out = np.empty(n, dtype=np.float64)
    check_opt(_lib.load().lmcma_host_gauss_stream(int(seed), n, _dp(out)), "lmcma_host_gauss_stream")
    return out


def uniform_stream(seed, n):
    out = np.empty(n, dtype=np.float64)
    check_opt(_lib.load().lmcma_host_uniform_stream(int(seed), n, _dp(out)), "lmcma_host_uniform_stream")
    return out


def covariance(num_dims, num_waypoints):
    """covariance() of lmcma.cpp:769-810, returned as an (n, n) array."""
    n = num_dims * num_waypoints
    out = np.empty((n, n), dtype=np.float64)
    check_opt(_lib.load().lmcma_host_covariance(num_dims, num_waypoints, _dp(out)), "lmcma_host_covariance")
    return out


def differentiationMatrix(num_time_steps, order, dt=1.0):
    out = np.zeros((num_time_steps, num_time_steps), dtype=np.float64)
    check_opt(_lib.load().lmcma_host_differentiation_matrix(num_time_steps, order, dt, _dp(out), -1),
              "lmcma_host_differentiation_matrix")
    return out


def cholesky(C):
    C = _f64(C)
    n = C.shape[0]
    L = np.empty((n, n), dtype=np.float64)
    check_opt(_lib.load().lmcma_host_cholesky(_dp(C), _dp(L), n), "lmcma_host_cholesky")
    return L  # column-major lower factor: L[c, r] = factor(r, c)


def applyCovL(L, z):
    L, z = _f64(L), _f64(z).copy()
    check_opt(_lib.load().lmcma_host_apply_cov_l(_dp(L), _dp(z), z.size), "lmcma_host_apply_cov_l")
    return z


def stable_sort(values):
    v = _f64(values).copy()
    idx = np.empty(v.size, dtype=np.int32)
    check_opt(_lib.load().lmcma_host_sort(v.size, _dp(v), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
              "lmcma_host_sort")
    return v, idx
