"""Thin numpy-facing wrapper over the C ABI of include/lmcma_hip.h.

One :class:`HipContext` = one ``lmcma_hip_ctx``: a HIP stream plus the device
buffers of (a shard of) one population.  Method names follow the C entry
points; see the header for which reference code each one replaces.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (HipConfig, OptSummary, c_double_p, c_float_p, c_int32_p, c_int64_p, check_hip)

Z_HOST, Z_PHILOX = 0, 1
GRID_BITS, GRID_U8, GRID_F32 = 0, 1, 2
PAD_ENDPOINT, PAD_ZERO = 0, 1
COST_TRAJECTORY, COST_EXTERNAL = 0, 1
(BUF_ROWS_LOCAL, BUF_COST_LOCAL, BUF_COST_ALL, BUF_PARTIAL_LOCAL, BUF_PARTIAL_ALL, BUF_XMEAN,
 BUF_SORTED_COST, BUF_PERM) = range(8)
KERNEL_CLASSES = ("sample", "cost", "select", "recombine", "update", "generate")


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class HipContext:
    def __init__(self, n_params, lambda_total, m, dims=0, waypoints=0, lambda_local=None,
                 first_candidate=0, mu=None, z_source=Z_PHILOX, cost_source=None, pad_mode=PAD_ENDPOINT,
                 seed=42, w_len=1.0, w_grid=1000.0, w_smooth=1.0, start=(0, 0, 0), goal=(0, 0, 0),
                 device=0, stream=None, queries=1):
        self.lib = _lib.load()
        cfg = HipConfig()
        cfg.struct_size = ctypes.sizeof(HipConfig)
        cfg.device = device
        cfg.dims = dims
        cfg.waypoints = waypoints
        cfg.n_params = n_params
        cfg.lambda_total = lambda_total
        cfg.lambda_local = lambda_total if lambda_local is None else lambda_local
        cfg.first_candidate = first_candidate
        cfg.m = m
        cfg.mu = lambda_total // 2 if mu is None else mu
        cfg.z_source = z_source
        if cost_source is None:
            cost_source = COST_TRAJECTORY if dims > 0 else COST_EXTERNAL
        cfg.cost_source = cost_source
        cfg.pad_mode = pad_mode
        cfg.seed = seed
        cfg.w_len, cfg.w_grid, cfg.w_smooth = w_len, w_grid, w_smooth
        s3 = list(start) + [0.0] * (3 - len(start))
        g3 = list(goal) + [0.0] * (3 - len(goal))
        for d in range(3):
            cfg.start[d] = float(s3[d])
            cfg.goal[d] = float(g3[d])
        cfg.stream = stream
        cfg.queries = queries
        self.queries = queries
        self.cfg = cfg
        self.N = n_params
        self.lam = lambda_total
        self.nloc = cfg.lambda_local
        self.m = m
        self.mu = cfg.mu
        self._h = ctypes.c_void_p()
        check_hip(self.lib.lmcma_hip_create(ctypes.byref(cfg), ctypes.byref(self._h)), "lmcma_hip_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.lmcma_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs -------------------------------------------------------------------
    def set_grid(self, cells, cell_type=None, c_obs=1.0):
        cells = np.ascontiguousarray(cells)
        if cell_type is None:
            cell_type = {np.dtype(np.uint8): GRID_U8, np.dtype(np.float32): GRID_F32,
                         np.dtype(np.uint32): GRID_BITS}[cells.dtype]
        if cell_type == GRID_BITS:
            raise ValueError("pass the extent explicitly with set_grid_bits for packed words")
        shape = cells.shape
        nz, ny, nx = (1,) * (3 - len(shape)) + tuple(shape)
        self._grid_keep = cells
        check_hip(self.lib.lmcma_hip_set_grid(self._h, cells.ctypes.data_as(ctypes.c_void_p), nx, ny, nz,
                                              cell_type, float(c_obs)), "lmcma_hip_set_grid")

    def set_grid_bits(self, words, nx, ny, nz=1, c_obs=1.0):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        check_hip(self.lib.lmcma_hip_set_grid(self._h, words.ctypes.data_as(ctypes.c_void_p), nx, ny, nz,
                                              GRID_BITS, float(c_obs)), "lmcma_hip_set_grid")

    def set_grid_from_occupancy(self, occ, d_min=0.5, want_distances=False):
        """Builds the clearance cost map 1 / max(distance, d_min) on the device from an occupancy
        array shaped (nx,), (ny, nx) or (nz, ny, nx); returns the distances when asked for."""
        occ = np.ascontiguousarray(np.asarray(occ) != 0, dtype=np.uint8)
        shape = (1,) * (3 - occ.ndim) + occ.shape
        nz, ny, nx = shape
        dist = np.empty(shape, dtype=np.float32) if want_distances else None
        check_hip(self.lib.lmcma_hip_set_grid_from_occupancy(
            self._h, occ.ctypes.data_as(ctypes.c_void_p), nx, ny, nz, float(d_min),
            dist.ctypes.data_as(ctypes.c_void_p) if want_distances else None), "lmcma_hip_set_grid_from_occupancy")
        return dist.reshape(occ.shape) if want_distances else None

    def set_bounds(self, lo, hi):
        lo, hi = _f64(lo), _f64(hi)
        check_hip(self.lib.lmcma_hip_set_bounds(self._h, _dp(lo), _dp(hi)), "lmcma_hip_set_bounds")

    def set_cov_factor(self, L):
        """Lower Cholesky factor of the sampling covariance, column-major [N*N] as the
        reference's cholesky() stores it (or an (N, N) array holding Lm transposed, which is
        the same memory); None for the identity."""
        L = _f64(L)
        if L is not None and L.size != self.N * self.N:
            raise ValueError("set_cov_factor: the factor has N * N entries")
        check_hip(self.lib.lmcma_hip_set_cov_factor(self._h, _dp(L)), "lmcma_hip_set_cov_factor")

    def set_state(self, xmean, sigma, V=None, PC=None, nj=None, M=1.0):
        xmean, V, PC, nj = _f64(xmean), _f64(V), _f64(PC), _f64(nj)
        k = 0 if V is None else V.shape[0]
        check_hip(self.lib.lmcma_hip_set_state(self._h, _dp(xmean), float(sigma), _dp(V), _dp(PC), _dp(nj), k,
                                               float(M)), "lmcma_hip_set_state")

    # ---- population steps -----------------------------------------------------------
    def sample(self, generation=0, z=None):
        z = _f64(z)
        check_hip(self.lib.lmcma_hip_sample(self._h, int(generation), _dp(z)), "lmcma_hip_sample")

    def evaluate(self):
        check_hip(self.lib.lmcma_hip_evaluate(self._h), "lmcma_hip_evaluate")

    def set_costs(self, costs, first=0):
        costs = _f64(costs)
        check_hip(self.lib.lmcma_hip_set_costs(self._h, first, costs.size, _dp(costs)), "lmcma_hip_set_costs")

    def select(self):
        sc = np.empty(self.lam, dtype=np.float32)
        perm = np.empty(self.lam, dtype=np.int32)
        check_hip(self.lib.lmcma_hip_select(self._h, sc.ctypes.data_as(c_float_p), perm.ctypes.data_as(c_int32_p)),
                  "lmcma_hip_select")
        return sc, perm

    def recombine(self, weights=None):
        weights = _f64(weights)
        out = np.empty(self.N, dtype=np.float64)
        check_hip(self.lib.lmcma_hip_recombine(self._h, _dp(weights), _dp(out)), "lmcma_hip_recombine")
        return out

    def rows(self, first=0, count=None):
        count = self.nloc - first if count is None else count
        out = np.empty((count, self.N), dtype=np.float64)
        check_hip(self.lib.lmcma_hip_get_rows(self._h, first, count, _dp(out)), "lmcma_hip_get_rows")
        return out

    def costs(self, first=0, count=None):
        count = self.nloc - first if count is None else count
        out = np.empty(count, dtype=np.float64)
        check_hip(self.lib.lmcma_hip_get_costs(self._h, first, count, _dp(out)), "lmcma_hip_get_costs")
        return out

    def terms(self, first=0, count=None):
        count = self.nloc - first if count is None else count
        out = np.empty((count, 3), dtype=np.float64)
        check_hip(self.lib.lmcma_hip_get_terms(self._h, first, count, _dp(out)), "lmcma_hip_get_terms")
        return out

    def score_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        count = rows.shape[0]
        cost = np.empty(count, dtype=np.float32)
        terms = np.empty((count, 3), dtype=np.float32)
        check_hip(self.lib.lmcma_hip_score_rows(self._h, rows.ctypes.data_as(c_float_p), count,
                                                cost.ctypes.data_as(c_float_p), terms.ctypes.data_as(c_float_p)),
                  "lmcma_hip_score_rows")
        return cost, terms

    # ---- device-resident optimiser ------------------------------------------------------
    def opt_init(self, xmean0, sigma0):
        xmean0 = _f64(xmean0)
        check_hip(self.lib.lmcma_hip_opt_init(self._h, _dp(xmean0), float(sigma0)), "lmcma_hip_opt_init")

    def opt_set_query(self, query, xmean0, sigma0):
        """Query `query` of a context with several queries starts from its own mean and step size."""
        xmean0 = _f64(xmean0)
        check_hip(self.lib.lmcma_hip_opt_set_query(self._h, int(query), _dp(xmean0), float(sigma0)),
                  "lmcma_hip_opt_set_query")

    def query_select(self, query):
        """The readers (rows, costs, terms, opt_get, opt_vectors, buffer) address this query from now on."""
        check_hip(self.lib.lmcma_hip_query_select(self._h, int(query)), "lmcma_hip_query_select")

    def opt_reset(self):
        check_hip(self.lib.lmcma_hip_opt_reset(self._h), "lmcma_hip_opt_reset")

    def run(self, generations):
        check_hip(self.lib.lmcma_hip_run(self._h, int(generations)), "lmcma_hip_run")

    def phase_local(self):
        check_hip(self.lib.lmcma_hip_phase_local(self._h), "lmcma_hip_phase_local")

    def phase_local_range(self, first, count):
        """Sample + score candidates [first, first + count) of a context that holds a whole population."""
        check_hip(self.lib.lmcma_hip_phase_local_range(self._h, int(first), int(count)), "lmcma_hip_phase_local_range")

    def phase_rank(self):
        check_hip(self.lib.lmcma_hip_phase_rank(self._h), "lmcma_hip_phase_rank")

    def phase_update(self, n_partials=0):
        check_hip(self.lib.lmcma_hip_phase_update(self._h, int(n_partials)), "lmcma_hip_phase_update")

    # ---- a population sharded over processes: RCCL inside the library ----------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId; one process makes them, every rank passes them to comm_init."""
        buf = ctypes.create_string_buffer(128)
        check_hip(_lib.load().lmcma_hip_comm_unique_id(buf), "lmcma_hip_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        if len(unique_id) != 128:
            raise ValueError("the unique id has 128 bytes")
        check_hip(self.lib.lmcma_hip_comm_init(self._h, bytes(unique_id), int(world), int(rank)),
                  "lmcma_hip_comm_init")

    def comm_attach(self, nccl_comm, world, rank):
        check_hip(self.lib.lmcma_hip_comm_attach(self._h, ctypes.c_void_p(nccl_comm), int(world), int(rank)),
                  "lmcma_hip_comm_attach")

    def comm_destroy(self):
        check_hip(self.lib.lmcma_hip_comm_destroy(self._h), "lmcma_hip_comm_destroy")

    def run_sharded(self, generations):
        check_hip(self.lib.lmcma_hip_run_sharded(self._h, int(generations)), "lmcma_hip_run_sharded")

    # ---- the same loop with the direct exchange (mailboxes the shards write into) -----------
    def peer_handle(self):
        """128 bytes that let the other shards map this context's mailbox."""
        buf = ctypes.create_string_buffer(128)
        check_hip(self.lib.lmcma_hip_peer_handle(self._h, buf), "lmcma_hip_peer_handle")
        return buf.raw

    def peer_connect(self, handles, rank):
        """handles: the peer_handle() of every shard, in rank order (this context's included)."""
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != 128 * len(handles):
            raise ValueError("every handle has 128 bytes")
        check_hip(self.lib.lmcma_hip_peer_connect(self._h, blob, len(handles), int(rank)), "lmcma_hip_peer_connect")

    def peer_enable(self, on=True):
        """False: run_sharded uses the communicator while the mailboxes stay mapped; True: back."""
        check_hip(self.lib.lmcma_hip_peer_enable(self._h, 1 if on else 0), "lmcma_hip_peer_enable")

    def peer_disconnect(self):
        check_hip(self.lib.lmcma_hip_peer_disconnect(self._h), "lmcma_hip_peer_disconnect")

    def peer_status(self):
        fine, err = ctypes.c_int(0), ctypes.c_int(0)
        check_hip(self.lib.lmcma_hip_peer_status(self._h, ctypes.byref(fine), ctypes.byref(err)), "lmcma_hip_peer_status")
        out = dict(fine_grained=bool(fine.value), wait_error=err.value)
        if err.value:
            out["detail"] = self.lib.lmcma_hip_last_error().decode()
        return out

    @staticmethod
    def run_sharded_group(contexts, generations):
        """The shards of one population held by this process (contexts[r] = shard r): the loop with
        the exchanges as device copies ordered by events."""
        arr = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
        check_hip(_lib.load().lmcma_hip_run_sharded_group(arr, len(contexts), int(generations)),
                  "lmcma_hip_run_sharded_group")

    def opt_get(self):
        sm = OptSummary()
        xm = np.empty(self.N, dtype=np.float64)
        check_hip(self.lib.lmcma_hip_opt_get(self._h, ctypes.byref(sm), _dp(xm)), "lmcma_hip_opt_get")
        return dict(itr=sm.itr, iterator_sz=sm.iterator_sz, counteval=sm.counteval, best_f=sm.best_f,
                    sigma=sm.sigma, s=sm.s, done=bool(sm.done), xmean=xm)

    def opt_vectors(self):
        N, m = self.N, self.m
        pc = np.empty(N)
        v = np.empty((m, N))
        p = np.empty((m, N))
        nj = np.empty(m)
        lj = np.empty(m)
        it = np.empty(m, dtype=np.int32)
        check_hip(self.lib.lmcma_hip_opt_get_vectors(self._h, _dp(pc), _dp(v), _dp(p), _dp(nj), _dp(lj),
                                                     it.ctypes.data_as(c_int32_p)), "lmcma_hip_opt_get_vectors")
        return dict(pc=pc, v_arr=v, pc_arr=p, nj=nj, lj=lj, iterator=it)

    def buffer(self, which):
        ptr = ctypes.c_void_p()
        nbytes = ctypes.c_size_t()
        check_hip(self.lib.lmcma_hip_buffer(self._h, which, ctypes.byref(ptr), ctypes.byref(nbytes)),
                  "lmcma_hip_buffer")
        return ptr.value, nbytes.value

    def row_stride(self):
        return self.lib.lmcma_hip_row_stride(self._h)

    def synchronize(self):
        check_hip(self.lib.lmcma_hip_synchronize(self._h), "lmcma_hip_synchronize")

    def stream_mark(self, which):
        check_hip(self.lib.lmcma_hip_stream_mark(self._h, int(which)), "lmcma_hip_stream_mark")

    def stream_elapsed(self):
        """Milliseconds of device time between stream_mark(0) and stream_mark(1)."""
        ms = ctypes.c_double(0.0)
        check_hip(self.lib.lmcma_hip_stream_elapsed(self._h, ctypes.byref(ms)), "lmcma_hip_stream_elapsed")
        return ms.value

    def timing_enable(self, on=True):
        check_hip(self.lib.lmcma_hip_timing_enable(self._h, 1 if on else 0), "lmcma_hip_timing_enable")

    def timing_get(self):
        ms = np.zeros(len(KERNEL_CLASSES), dtype=np.float64)
        n = np.zeros(len(KERNEL_CLASSES), dtype=np.int64)
        check_hip(self.lib.lmcma_hip_timing_get(self._h, _dp(ms), n.ctypes.data_as(c_int64_p)),
                  "lmcma_hip_timing_get")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KERNEL_CLASSES)}

    def timing_event_overhead(self, pairs=512):
        """Milliseconds an empty event bracket reads on the context's stream."""
        ms = ctypes.c_double(0.0)
        check_hip(self.lib.lmcma_hip_timing_event_overhead(self._h, pairs, ctypes.byref(ms)),
                  "lmcma_hip_timing_event_overhead")
        return ms.value
