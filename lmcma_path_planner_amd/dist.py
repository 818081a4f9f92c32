"""One population sharded over the GPUs of a node: one process per GPU,
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI) for the exchange steps.

The reference is single-process (no collective anywhere in it); the partition is:
rank g owns candidates [g*lambda/G, (g+1)*lambda/G).  The Philox counter is the
global candidate index, so a candidate's row does not depend on G.  Per generation:

  1. phase_local   sample + score the own candidates           (no communication)
  2. all-gather    the own costs -> every rank holds all lambda costs
  3. phase_rank    every rank finds the ranks of its own candidates among all costs
                   and sums the weighted rows of the elites it owns; with its share
                   of the success-rule count and the lowest cost -> one row of N + 2
  4. all-gather    the per-rank rows (G x (N + 2) doubles)
  5. phase_update  every rank adds the G rows in rank order and runs the LM-CMA
                   update; the replicated state stays bit-identical

The elite exchange moves each rank's weighted elite sum rather than the elite
rows themselves: mu = lambda/2 (reference lmcma.cpp:136), so "the rows of the
elites" is on average half of every shard, while their weighted sum is N doubles.

The other form, ``gathered_generation``, does move the rows: every rank holds the whole
population, samples its range, all-gathers rows and costs, and ranks and recombines the
whole population itself (reference lmcma.cpp:313-329 over all of arx).  Its state is
that of one unsharded context bit for bit, whatever the number of ranks, at the price of
lambda * row stride * 4 bytes received per rank and generation.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import hip as hipmod


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch
    can alias a buffer the library owns (no copy)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def alias_tensor(ctx, which, dtype, device):
    ptr, nbytes = ctx.buffer(which)
    itemsize = torch.empty((), dtype=dtype).element_size()
    typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DeviceArray(ptr, nbytes // itemsize, typestr), device=device)


class HipShard:
    """A HipContext plus torch views of the buffers the collectives move."""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.cost_all = alias_tensor(ctx, hipmod.BUF_COST_ALL, torch.float32, device)
        self.cost_local = alias_tensor(ctx, hipmod.BUF_COST_LOCAL, torch.float32, device)
        self.partial_local = alias_tensor(ctx, hipmod.BUF_PARTIAL_LOCAL, torch.float64, device)
        self.partial_all = alias_tensor(ctx, hipmod.BUF_PARTIAL_ALL, torch.float64, device)
        self.N = ctx.N
        self.partial_len = self.partial_local.numel()  # N + 2: elite sum, success-rule count, lowest cost

    def phase_local(self):
        self.ctx.phase_local()

    def phase_rank(self):
        self.ctx.phase_rank()

    def phase_update(self, n_partials):
        self.ctx.phase_update(n_partials)


class HipWhole:
    """A HipContext that holds the whole population (one rank of the rows exchange) plus torch
    views of the two buffers whose other ranges arrive by all-gather."""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.ld = ctx.row_stride()
        self.rows_all = alias_tensor(ctx, hipmod.BUF_ROWS_LOCAL, torch.float32, device)   # [lambda][ld]
        self.cost_all = alias_tensor(ctx, hipmod.BUF_COST_ALL, torch.float32, device)
        self.lam = self.cost_all.numel()

    def phase_local_range(self, first, count):
        self.ctx.phase_local_range(first, count)

    def phase_rank(self):
        self.ctx.phase_rank()

    def phase_update(self, n_partials):
        self.ctx.phase_update(n_partials)


def shard_range(lam_total, world, rank):
    """Contiguous equal shards; lambda must divide evenly (all-gather needs equal blocks)."""
    if lam_total % world:
        raise ValueError("lambda_total must be a multiple of the number of ranks")
    per = lam_total // world
    return rank * per, per


def _all_gather(out, inp, group):
    """all_gather_into_tensor; device tensors go straight to RCCL, and through a host
    copy on any other backend (gloo is only used to rehearse the loop without RCCL)."""
    if inp.is_cuda and dist.get_backend(group) != "nccl":
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def sharded_generation(shard, group=None):
    """One generation of a sharded population.  `shard` provides phase_local /
    phase_rank / phase_update and the tensors cost_local, cost_all, partial_local,
    partial_all (device tensors with nccl, CPU tensors with gloo).

    The library's kernels and the collectives must be ordered on one stream: create
    the HipContext on torch's current stream (``stream=torch.cuda.current_stream().cuda_stream``
    of a non-default stream) and call this function with that stream current."""
    world = dist.get_world_size(group)
    shard.phase_local()
    if world == 1:
        shard.phase_rank()
        shard.phase_update(0)
        return
    # cost_local is the rank's own block inside cost_all, so this is an in-place all-gather
    _all_gather(shard.cost_all, shard.cost_local, group)
    shard.phase_rank()
    _all_gather(shard.partial_all[: world * shard.partial_len], shard.partial_local, group)
    shard.phase_update(world)


def gathered_generation(whole, group=None):
    """One generation of the rows exchange.  `whole` provides phase_local_range / phase_rank /
    phase_update, the row stride ld, and the tensors rows_all ([lambda * ld]) and cost_all
    ([lambda]) of a context that holds the whole population.  Stream rule as above."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = shard_range(whole.lam, world, rank)
    whole.phase_local_range(first, count)
    if world > 1:
        ld = whole.ld
        # both in place: the rank's own range lies at its position inside the buffers
        _all_gather(whole.rows_all[: whole.lam * ld], whole.rows_all[first * ld:(first + count) * ld], group)
        _all_gather(whole.cost_all, whole.cost_all[first:first + count], group)
    whole.phase_rank()
    whole.phase_update(0)
