"""Synthetic cost maps and the benchmark problem set (host side).

Grids are indexed ``[z][y][x]`` (row = y, col = x) like the reference's
``EDT_Matrix(M=y, N=x)`` lookups
(sample_based_optimisation_based_path_planner.cpp:599-602).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import c_double_p, c_float_p, c_int32_p, c_uint8_p


def _u8p(a):
    return a.ctypes.data_as(c_uint8_p)


def two_bar_map(size, nz=1):
    """The reference's two-bar map (…path_planner.cpp:269-297) scaled to size^2 (x nz)."""
    occ = np.zeros((nz, size, size), dtype=np.uint8)
    if _lib.load().lmcma_grid_two_bar(size, nz, _u8p(occ)) != 0:
        raise ValueError("lmcma_grid_two_bar: bad arguments")
    return occ if nz > 1 else occ[0]


def add_random_boxes(occ, seed, fill, start, goal):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    shape = occ.shape
    nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
    s = np.asarray(list(start) + [0.0] * (3 - len(start)), dtype=np.float64)
    g = np.asarray(list(goal) + [0.0] * (3 - len(goal)), dtype=np.float64)
    if _lib.load().lmcma_grid_random_boxes(nx, ny, nz, int(seed), float(fill), s.ctypes.data_as(c_double_p),
                                           g.ctypes.data_as(c_double_p), _u8p(occ)) != 0:
        raise ValueError("lmcma_grid_random_boxes: bad arguments")
    return occ


def edt(occ):
    """Exact Euclidean distance (cells) to the nearest occupied cell."""
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    shape = occ.shape
    nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
    dist = np.empty(shape, dtype=np.float32)
    if _lib.load().lmcma_grid_edt(_u8p(occ), nx, ny, nz, dist.ctypes.data_as(c_float_p)) != 0:
        raise ValueError("lmcma_grid_edt: bad arguments")
    return dist


def ssedt8(inside):
    """8SSEDT of …path_planner.cpp:447-490: (signed int distance, distance to marked cells)."""
    inside = np.ascontiguousarray(inside, dtype=np.uint8)
    h, w = inside.shape
    sd = np.empty((h, w), dtype=np.int32)
    d = np.empty((h, w), dtype=np.float64)
    if _lib.load().lmcma_grid_8ssedt(_u8p(inside), w, h, sd.ctypes.data_as(c_int32_p),
                                     d.ctypes.data_as(c_double_p)) != 0:
        raise ValueError("lmcma_grid_8ssedt: bad arguments")
    return sd, d


def read_binvox(path):
    """(occ[z][y][x] uint8, translate[3], scale) of a binvox voxel file."""
    dims = (ctypes.c_int * 3)()
    tr = np.zeros(3)
    sc = ctypes.c_double(1.0)
    lib = _lib.load()
    rc = lib.lmcma_grid_read_binvox(os.fsencode(path), dims, tr.ctypes.data_as(c_double_p), ctypes.byref(sc), None, 0)
    if rc < 0:
        raise ValueError("read_binvox(%s): error %d" % (path, rc))
    nx, ny, nz = dims[0], dims[1], dims[2]
    occ = np.zeros((nz, ny, nx), dtype=np.uint8)
    rc = lib.lmcma_grid_read_binvox(os.fsencode(path), dims, tr.ctypes.data_as(c_double_p), ctypes.byref(sc), _u8p(occ),
                                    occ.size)
    if rc < 0:
        raise ValueError("read_binvox(%s): error %d" % (path, rc))
    return occ, tr, sc.value


def write_binvox(path, occ, translate=(0.0, 0.0, 0.0), scale=1.0):
    """Writes occ[z][y][x] as a binvox file (voxel order x-major, then z, then y; runs of at most 255)."""
    occ = np.asarray(occ)
    nz, ny, nx = occ.shape
    flat = (occ.transpose(2, 0, 1) != 0).astype(np.uint8).reshape(-1)     # [x][z][y]
    edges = np.flatnonzero(np.diff(flat)) + 1
    starts = np.concatenate([[0], edges])
    lengths = np.diff(np.concatenate([starts, [flat.size]]))
    with open(path, "wb") as f:
        f.write(("#binvox 1\ndim %d %d %d\ntranslate %.17g %.17g %.17g\nscale %.17g\ndata\n"
                 % ((nx, nz, ny) + tuple(translate) + (scale,))).encode())
        out = bytearray()
        for st, ln in zip(starts.tolist(), lengths.tolist()):
            v = int(flat[st])
            while ln > 0:
                c = min(ln, 255)
                out += bytes((v, c))
                ln -= c
        f.write(bytes(out))


def read_bmp(path):
    """Pixels of a BMP image with green < 128, as a (height, width) uint8 array, top row first."""
    w, h = ctypes.c_int(0), ctypes.c_int(0)
    lib = _lib.load()
    rc = lib.lmcma_grid_read_bmp(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), None, 0)
    if rc < 0:
        raise ValueError("read_bmp(%s): error %d" % (path, rc))
    marked = np.zeros((h.value, w.value), dtype=np.uint8)
    rc = lib.lmcma_grid_read_bmp(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), _u8p(marked), marked.size)
    if rc < 0:
        raise ValueError("read_bmp(%s): error %d" % (path, rc))
    return marked


def state_cost_from_distance(dist, d_min=0.5):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    cost = np.empty_like(dist)
    if _lib.load().lmcma_grid_state_cost(dist.ctypes.data_as(c_float_p), dist.size, float(d_min),
                                         cost.ctypes.data_as(c_float_p)) != 0:
        raise ValueError("lmcma_grid_state_cost: bad arguments")
    return cost


class Problem:
    """One benchmark configuration: grid, endpoints, initial mean, bounds, sizes."""

    def __init__(self, name, dims, size, waypoints, lam, m, esdf=False, fill=0.15, seed=0xC0FFEE):
        self.name, self.dims, self.size, self.W, self.lam, self.m, self.esdf = name, dims, size, waypoints, lam, m, esdf
        self.N = dims * waypoints
        nz = size if dims == 3 else 1
        hi = float(size - 1)
        # start (size-1, 0[, 0]) -> goal (0, size-1[, size-1]): the reference's (99,0) -> (0,99)
        # (…path_planner.cpp:702-706) scaled; bounds [0, size-1] (:697)
        self.start = [hi, 0.0] + ([0.0] if dims == 3 else [])
        self.goal = [0.0, hi] + ([hi] if dims == 3 else [])
        occ = two_bar_map(size, nz)
        occ = add_random_boxes(occ, seed, fill, self.start, self.goal)
        self.occ = occ
        if esdf:
            self.cells = state_cost_from_distance(edt(occ), 0.5)
        else:
            self.cells = occ
        self.lo = np.zeros(self.N)
        self.hi = np.full(self.N, hi)
        # straight line start -> goal, interior points, dimension-major
        t = (np.arange(1, waypoints + 1, dtype=np.float64)) / (waypoints + 1)
        self.x0 = np.concatenate([self.start[d] + t * (self.goal[d] - self.start[d]) for d in range(dims)])
        self.sigma0 = size / 20.0
        self.weights = (1.0, 1000.0, 1.0)  # length, grid, smoothness ("longsafe", :684-690)

    def grid_bytes(self):
        return self.cells.size * 4 if self.esdf else (self.cells.size + 7) // 8

    def algorithmic_bytes_per_candidate(self, lam_per_gpu=None):
        """HBM bytes one candidate-trajectory needs if every operand moves once."""
        lam = self.lam if lam_per_gpu is None else lam_per_gpu
        N = self.N
        return 4 * (2 * N + 1) + 4 * (N // 2) + 12 + (self.grid_bytes() + 8 * self.m * N) / lam


def benchmark_problem(cfg):
    """The five configurations of the headline benchmark (1-based)."""
    table = {
        1: dict(name="cfg1-2d100-w32", dims=2, size=100, waypoints=32, lam=16, m=4),
        2: dict(name="cfg2-2d512-w64", dims=2, size=512, waypoints=64, lam=4096, m=16),
        3: dict(name="cfg3-3d128-w48", dims=3, size=128, waypoints=48, lam=8192, m=32),
        4: dict(name="cfg4-2d1024-w128", dims=2, size=1024, waypoints=128, lam=65536, m=32),
        5: dict(name="cfg5-3d256-esdf-w96", dims=3, size=256, waypoints=96, lam=131072, m=32, esdf=True),
    }
    return Problem(seed=0xC0FFEE + cfg, **table[cfg])
