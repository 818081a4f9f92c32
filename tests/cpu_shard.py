"""CPU stand-in for one rank's shard, used only by the gloo tests of the sharded
generation loop (lmcma_path_planner_amd/dist.py).  It mirrors the semantics of the
device phases with the oracle's fp64 functions: Philox rows by global candidate
index, fp32-rounded rows and costs, stable ranking of all costs, rank-weighted
partial sums of the own elites, replicated update."""
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
        self.partial_local = torch.zeros(prob.N, dtype=torch.float64)
        self.partial_all = torch.zeros(max_partials * prob.N, dtype=torch.float64)
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
        self.cost_local.copy_(torch.from_numpy(cost.astype(np.float32)))

    def phase_rank(self):
        costs = self.cost_all.numpy().astype(np.float64)
        self.order = np.argsort(costs, kind="stable")
        w = self.o.weights
        acc = np.zeros(self.N)
        for r in range(self.o.mu):
            local = self.order[r] - self.first
            if 0 <= local < self.nloc:
                acc += w[r] * self.rows[local].astype(np.float64)
        self.partial_local.copy_(torch.from_numpy(acc))

    def phase_update(self, n_partials):
        if n_partials == 0:
            xnew = self.partial_local.numpy().copy()
        else:
            parts = self.partial_all.numpy()[: n_partials * self.N].reshape(n_partials, self.N)
            xnew = np.zeros(self.N)
            for p in range(n_partials):
                xnew += parts[p]
        self.o.update_with_mean(self.cost_all.numpy().astype(np.float64), xnew)
