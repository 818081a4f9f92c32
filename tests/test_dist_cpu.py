"""The sharded generation loop (one population over several ranks) on CPU:
world_size 2 and 4 under gloo against the unsharded run.  The ranks run
lmcma_path_planner_amd.dist.sharded_generation unchanged; the device phases are
replaced by the CPU stand-in of tests/cpu_shard.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENS = 6
LAM = 32


def _problem():
    from lmcma_path_planner_amd import grids
    prob = grids.benchmark_problem(1)
    prob.lam = LAM
    return prob


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_shard import CpuShard
    from lmcma_path_planner_amd.dist import shard_range, sharded_generation
    prob = _problem()
    first, nloc = shard_range(LAM, world, rank)
    shard = CpuShard(prob, LAM, first, nloc, seed=42)
    for _ in range(GENS):
        sharded_generation(shard)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), xmean=shard.o.xmean, sigma=shard.o.sigma,
             cost_all=shard.cost_all.numpy(), v=shard.o.v_arr, itr=shard.o.itr)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tmp_path, port):
    out = str(tmp_path / ("w%d" % world))
    os.makedirs(out, exist_ok=True)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_generation_matches_unsharded(world, tmp_path):
    single = _run(1, tmp_path, 29531)[0]
    ranks = _run(world, tmp_path, 29533 + world)
    assert int(single["itr"]) == GENS
    for r in ranks:
        assert int(r["itr"]) == GENS
        # every rank holds the same replicated state, bit for bit
        np.testing.assert_array_equal(r["xmean"], ranks[0]["xmean"])
        np.testing.assert_array_equal(r["v"], ranks[0]["v"])
        assert float(r["sigma"]) == float(ranks[0]["sigma"])
        # the gathered costs (each rank's block in ascending order) are the unsharded run's costs
        # exactly: rows do not depend on the sharding
        np.testing.assert_array_equal(np.sort(r["cost_all"]), np.sort(single["cost_all"]))
        nloc = LAM // world
        assert all(np.all(np.diff(r["cost_all"][g * nloc:(g + 1) * nloc]) >= 0) for g in range(world))
        # the mean differs from the unsharded run only by the grouping of the fp64 partial sums
        np.testing.assert_allclose(r["xmean"], single["xmean"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(float(r["sigma"]), float(single["sigma"]), rtol=1e-12)


def _rows_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_shard import CpuWhole
    from lmcma_path_planner_amd.dist import gathered_generation
    whole = CpuWhole(_problem(), LAM, seed=42)
    for _ in range(GENS):
        gathered_generation(whole)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), xmean=whole.o.xmean, sigma=whole.o.sigma,
             cost_all=whole.cost_all.numpy(), rows_all=whole.rows_all.numpy(), v=whole.o.v_arr, itr=whole.o.itr)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rows_exchange_is_the_unsharded_run_bit_for_bit(world, tmp_path):
    """dist.gathered_generation: every rank holds the whole population and samples its range; rows
    and costs are all-gathered in place.  State, rows and costs equal the one-rank run exactly."""
    def run(w, port):
        out = str(tmp_path / ("rows%d" % w))
        os.makedirs(out, exist_ok=True)
        mp.spawn(_rows_worker, args=(w, port, out), nprocs=w, join=True)
        return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(w)]
    single = run(1, 29561)[0]
    for r in run(world, 29563 + world):
        assert int(r["itr"]) == GENS
        for k in ("xmean", "v", "cost_all", "rows_all"):
            np.testing.assert_array_equal(r[k], single[k], err_msg=k)
        assert float(r["sigma"]) == float(single["sigma"])


def test_shard_range():
    from lmcma_path_planner_amd.dist import shard_range
    assert shard_range(4096 * 8, 8, 3) == (3 * 4096, 4096)
    with pytest.raises(ValueError):
        shard_range(100, 8, 0)
