"""The sharded generation loop inside the library (lmcma_hip_run_sharded: RCCL all-gathers
queued between the phases).  One GPU can hold one rank only (RCCL refuses two ranks on one
device), so what runs here is a one-rank communicator: every call the loop makes on a node
of G GPUs, with G = 1.  The orchestration for G > 1 is covered by the gloo tests on the CPU.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def make_ctx(prob, lam, **kw):
    from lmcma_path_planner_amd import hip
    ctx = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start,
                         goal=prob.goal, w_len=prob.weights[0], w_grid=prob.weights[1], w_smooth=prob.weights[2], **kw)
    ctx.set_grid(prob.cells)
    ctx.set_bounds(prob.lo, prob.hi)
    ctx.opt_init(prob.x0, prob.sigma0)
    return ctx


@pytest.mark.parametrize("cfg,lam,gens", [(1, 16, 30), (2, 1024, 40), (3, 2048, 44)])
def test_one_rank_communicator_gives_the_bits_of_run(cfg, lam, gens):
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(cfg)
    a = make_ctx(prob, lam)
    a.run(gens)
    b = make_ctx(prob, lam)
    b.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
    b.run_sharded(gens // 2)
    b.run_sharded(gens - gens // 2)
    sa, sb = a.opt_get(), b.opt_get()
    assert sa["itr"] == sb["itr"] == gens
    assert sa["sigma"] == sb["sigma"] and sa["best_f"] == sb["best_f"] and sa["counteval"] == sb["counteval"]
    np.testing.assert_array_equal(sa["xmean"], sb["xmean"])
    va, vb = a.opt_vectors(), b.opt_vectors()
    for k in va:
        np.testing.assert_array_equal(va[k], vb[k], err_msg=k)
    np.testing.assert_array_equal(a.rows(), b.rows())
    np.testing.assert_array_equal(a.costs(), b.costs())
    a.close()
    b.close()


def test_in_place_all_gather_leaves_the_own_block_alone():
    """The costs' all-gather sends from inside its receive buffer (the shard's own block)."""
    from lmcma_path_planner_amd import grids, hip
    prob = grids.benchmark_problem(2)
    c = make_ctx(prob, 512)
    c.phase_local()
    before = c.costs().copy()
    c.close()
    d = make_ctx(prob, 512)
    d.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
    d.run_sharded(1)
    np.testing.assert_array_equal(d.costs(), before)
    d.close()


def test_comm_errors():
    from lmcma_path_planner_amd import _lib, grids, hip
    prob = grids.benchmark_problem(1)
    c = make_ctx(prob, 16)
    with pytest.raises(_lib.LmcmaError):
        c.run_sharded(1)                       # no communicator
    uid = hip.HipContext.comm_unique_id()
    with pytest.raises(_lib.LmcmaError):
        c.comm_init(uid, 2, 0)                 # the context holds the whole population, not half of it
    with pytest.raises(_lib.LmcmaError):
        c.comm_init(uid, 1, 1)                 # rank outside the communicator
    c.comm_init(uid, 1, 0)
    with pytest.raises(_lib.LmcmaError):
        c.comm_init(uid, 1, 0)                 # already has one
    c.comm_destroy()
    c.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
    c.run_sharded(3)
    assert c.opt_get()["itr"] == 3
    c.close()
    # a shard of a larger population is refused by a communicator of the wrong size
    s = hip.HipContext(prob.N, 64, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=16, first_candidate=16,
                       seed=42, start=prob.start, goal=prob.goal)
    with pytest.raises(_lib.LmcmaError):
        s.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
    s.close()


GRAPH_SCRIPT = """
import sys
import numpy as np
sys.path.insert(0, %r)
from lmcma_path_planner_amd import grids, hip
prob = grids.benchmark_problem(2)
def make():
    c = hip.HipContext(prob.N, 1024, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start, goal=prob.goal)
    c.set_grid(prob.cells); c.set_bounds(prob.lo, prob.hi); c.opt_init(prob.x0, prob.sigma0)
    return c
a = make(); a.run(37)
b = make(); b.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
b.run_sharded(5); b.run_sharded(12); b.run_sharded(20)
sa, sb = a.opt_get(), b.opt_get()
assert sa["itr"] == sb["itr"] == 37 and sa["sigma"] == sb["sigma"] and sa["best_f"] == sb["best_f"]
assert np.array_equal(sa["xmean"], sb["xmean"])
print("replayed sharded generations: identical")
"""


def test_recorded_sharded_generations_give_the_same_bits():
    """LMCMA_HIP_SHARD_GRAPH=1: two sharded generations, collectives included, replayed from a HIP
    graph (the switch is read once per process, hence the child process)."""
    env = dict(os.environ, LMCMA_HIP_SHARD_GRAPH="1")
    out = subprocess.run([sys.executable, "-c", GRAPH_SCRIPT % ROOT], env=env, capture_output=True, text=True,
                         timeout=200)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical" in out.stdout


ROWS_SCRIPT = """
import sys
import numpy as np
sys.path.insert(0, %r)
from lmcma_path_planner_amd import grids, hip
for cfg, lam, gens in ((2, 4096, 30), (3, 2048, 40), (2, 32768, 12)):
    prob = grids.benchmark_problem(cfg)
    def make():
        c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, seed=42, start=prob.start, goal=prob.goal)
        c.set_grid(prob.cells); c.set_bounds(prob.lo, prob.hi); c.opt_init(prob.x0, prob.sigma0)
        return c
    a = make(); a.run(gens)
    b = make(); b.comm_init(hip.HipContext.comm_unique_id(), 1, 0)
    b.run_sharded(1); b.run_sharded(gens // 2); b.run_sharded(gens - 1 - gens // 2)
    sa, sb = a.opt_get(), b.opt_get()
    assert sa["itr"] == sb["itr"] == gens and sa["sigma"] == sb["sigma"] and sa["best_f"] == sb["best_f"], (cfg, lam)
    assert np.array_equal(sa["xmean"], sb["xmean"]) and np.array_equal(a.rows(), b.rows()) and np.array_equal(a.costs(), b.costs())
    a.close(); b.close()
print("rows exchange with one rank: identical")
"""


def test_rows_exchange_through_the_communicator_with_one_rank():
    """LMCMA_HIP_ROWS_EXCHANGE=1: a context that holds its whole population runs the rows form of
    lmcma_hip_run_sharded (its range = everything; the in-place all-gathers of rows and costs over a
    one-rank communicator) and ends in the bits of lmcma_hip_run.  With G > 1 the same code runs
    per rank; tests/test_gpu_sharded_multi.py executes G = 2, 4, 8 with device copies in place of RCCL."""
    env = dict(os.environ, LMCMA_HIP_ROWS_EXCHANGE="1")
    out = subprocess.run([sys.executable, "-c", ROWS_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical" in out.stdout


def test_class_in_shard_mode_with_one_rank_equals_the_plain_class():
    """LMCMA::setPopulationShard (world 1): step() goes through lmcma_hip_run_sharded."""
    from lmcma_path_planner_amd import grids, hip, planner
    prob = grids.benchmark_problem(1)

    def make(shard):
        o = planner.LMCMA(prob.x0, 64, prob.lo, prob.hi, sigma=prob.sigma0, inseed=42)
        o.setOptions(m=8, philox=True, device_update=True)
        if shard:
            o.setPopulationShard(1, 0, hip.HipContext.comm_unique_id())
        o.setTrajectoryProblem(prob.dims, prob.W, prob.cells, prob.start, prob.goal)
        o.init(prob.N)
        return o

    a, b = make(False), make(True)
    a.step(40)
    b.step(40)
    assert a.counteval == b.counteval == 40 * 64 and a.BestF == b.BestF
    np.testing.assert_array_equal(a.mean(), b.mean())
    a.close()
    b.close()
