"""CPU stand-in for one rank's shard, used only by the gloo tests of the sharded
generation loop (lmcma_path_planner_amd/dist.py).  It mirrors the semantics of the
device phases with the oracle's fp64 functions: Philox rows by global candidate
index, fp32-rounded rows and costs, the local costs exchanged in ascending order,
global ranks of the own candidates by merging the ranks' sorted lists (ties by rank
order, i.e. by candidate index), rank-weighted partial sums of the own elites,
replicated update."""
import numpy as np
import torch

from oracle import pyoracle


class CpuShard:
    def __init__(self, prob, lam_total, first, nloc, seed, max_partials=64):
        self.N, self.first, self.nloc, self.lam = prob.N, first, nloc, lam_total
        self.seed = seed
        self.co = pyoracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
        # every rank holds the whole (replicated) optimiser state
        self.o = pyoracle.OracleLMCMA(prob.N, lam_total, prob.m, initial=prob.x0, lo=prob.lo, hi=prob.hi,
                                      sigma=prob.sigma0, seed=seed, z_source=pyoracle.Z_GIVEN)
        self.cost_all = torch.zeros(lam_total, dtype=torch.float32)
        self.cost_local = self.cost_all[first:first + nloc]          # a view: in-place all-gather
        # a row is the elite sum plus two scalars (success-rule count, lowest cost), like the device's
        self.partial_len = prob.N + 2
        self.partial_local = torch.zeros(self.partial_len, dtype=torch.float64)
        self.partial_all = torch.zeros(max_partials * self.partial_len, dtype=torch.float64)
        self.rows = None
        self.order = None

    def phase_local(self):
        gen = self.o.itr
        Z = np.zeros((self.lam, self.N))
        for i in range(self.nloc):
            Z[self.first + i] = pyoracle.philox_normals(self.seed, gen, self.first + i, self.N)
        self.o.sample_given(Z)
        self.rows = self.o.arx[self.first:self.first + self.nloc].astype(np.float32)
        cost, _ = self.co.cost_rows_f32(self.rows)
        self.cost_raw = np.where(np.isnan(cost), np.inf, cost).astype(np.float32)
        self.local_order = np.argsort(self.cost_raw.astype(np.float64), kind="stable")
        self.cost_local.copy_(torch.from_numpy(self.cost_raw[self.local_order]))   # ascending

    def phase_rank(self):
        lists = self.cost_all.numpy().reshape(-1, self.nloc)
        me = self.first // self.nloc
        mine = lists[me]
        rank = np.arange(self.nloc)
        for g in range(lists.shape[0]):
            if g != me:
                rank = rank + np.searchsorted(lists[g], mine, side="right" if g < me else "left")
        w = self.o.weights
        acc = np.zeros(self.N)
        for p in np.argsort(rank):                      # own elites in rank order
            if rank[p] < self.o.mu:
                acc += w[rank[p]] * self.rows[self.local_order[p]].astype(np.float64)
        self.partial_local[: self.N].copy_(torch.from_numpy(acc))

    def phase_update(self, n_partials):
        if n_partials == 0:
            xnew = self.partial_local.numpy()[: self.N].copy()
        else:
            parts = self.partial_all.numpy()[: n_partials * self.partial_len].reshape(n_partials, self.partial_len)[:, : self.N]
            xnew = np.zeros(self.N)
            for p in range(n_partials):
                xnew += parts[p]
        self.o.update_with_mean(self.cost_all.numpy().astype(np.float64), xnew)


class CpuWhole:
    """Stand-in for one rank of the rows exchange (dist.gathered_generation): the whole population
    in every rank, the own range sampled and scored here, everything ranked and recombined here."""

    def __init__(self, prob, lam_total, seed):
        self.N, self.lam, self.seed = prob.N, lam_total, seed
        self.ld = prob.N
        self.co = pyoracle.CostOracle(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
        self.o = pyoracle.OracleLMCMA(prob.N, lam_total, prob.m, initial=prob.x0, lo=prob.lo, hi=prob.hi,
                                      sigma=prob.sigma0, seed=seed, z_source=pyoracle.Z_GIVEN)
        self.rows_all = torch.zeros(lam_total * self.ld, dtype=torch.float32)
        self.cost_all = torch.zeros(lam_total, dtype=torch.float32)

    def phase_local_range(self, first, count):
        gen = self.o.itr
        Z = np.zeros((self.lam, self.N))
        for i in range(first, first + count):
            Z[i] = pyoracle.philox_normals(self.seed, gen, i, self.N)
        self.o.sample_given(Z)
        rows = self.o.arx[first:first + count].astype(np.float32)
        cost, _ = self.co.cost_rows_f32(rows)
        self.rows_all[first * self.ld:(first + count) * self.ld].copy_(torch.from_numpy(rows.reshape(-1)))
        self.cost_all[first:first + count].copy_(torch.from_numpy(np.where(np.isnan(cost), np.inf, cost).astype(np.float32)))

    def phase_rank(self):
        cost = self.cost_all.numpy().astype(np.float64)
        order = np.argsort(cost, kind="stable")
        rows = self.rows_all.numpy().reshape(self.lam, self.ld)
        acc = np.zeros(self.N)
        for r in range(self.o.mu):                       # the mu best rows in rank order (lmcma.cpp:321-326)
            acc += self.o.weights[r] * rows[order[r]].astype(np.float64)
        self.xnew = acc

    def phase_update(self, n_partials):
        assert n_partials == 0
        self.o.update_with_mean(self.cost_all.numpy().astype(np.float64), self.xnew)
