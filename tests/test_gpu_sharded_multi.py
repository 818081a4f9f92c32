"""A population sharded over MORE THAN ONE rank, executed for real on the one GPU of the test box:

* lmcma_hip_run_sharded_group: G contexts of one process, exchanges as device copies ordered by events;
* lmcma_hip_run_sharded with the direct exchange (mailboxes + arrival counters), the G contexts in
  one process (plain pointers) and in G processes (IPC handles);

each against the phase-by-phase run of the same G contexts with the exchanges done by the test
(tests/test_gpu_baseline_shapes.py follows that form against the oracle at the stated shapes), bit for
bit: replicated state, direction vectors, rows and costs of every shard.  The loop forms carry the
update of a generation at the head of the next sampling launch, where it reads the G gathered rows;
the phase form runs it as a launch of its own.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SHAPES = {  # name: (benchmark problem, candidates per shard)
    "cfg2": (2, 1024),
    "cfg2_full_shard": (2, 4096),
    "cfg3": (3, 1024),
    "cfg1": (1, 64),
}


def make_shards(prob, G, nloc, seed=42, streams=None):
    from lmcma_path_planner_amd import hip
    ctxs = []
    for g in range(G):
        c = hip.HipContext(prob.N, G * nloc, prob.m, dims=prob.dims, waypoints=prob.W, lambda_local=nloc,
                           first_candidate=g * nloc, start=prob.start, goal=prob.goal, seed=seed, w_len=prob.weights[0],
                           w_grid=prob.weights[1], w_smooth=prob.weights[2])
        c.set_grid(prob.cells)
        c.set_bounds(prob.lo, prob.hi)
        c.opt_init(prob.x0, prob.sigma0)
        ctxs.append(c)
    return ctxs


def by_phases(prob, G, nloc, gens, seed=42):
    """The trusted form: every phase of every shard called from here, the exchanges as tensor copies."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    dev = torch.device("cuda", 0)
    ctxs = make_shards(prob, G, nloc, seed)
    shards = [shard_mod.HipShard(c, dev) for c in ctxs]
    L = shards[0].partial_len
    for _ in range(gens):
        for c in ctxs:
            c.phase_local()
        for c in ctxs:
            c.synchronize()
        for s in shards:
            for g, src in enumerate(shards):
                s.cost_all[g * nloc:(g + 1) * nloc].copy_(src.cost_local)
        torch.cuda.synchronize()
        for c in ctxs:
            c.phase_rank()
        for c in ctxs:
            c.synchronize()
        for s in shards:
            for g, src in enumerate(shards):
                s.partial_all[g * L:(g + 1) * L].copy_(src.partial_local)
        torch.cuda.synchronize()
        for c in ctxs:
            c.phase_update(G)
    out = snapshot(ctxs)
    for c in ctxs:
        c.close()
    return out


def snapshot(ctxs):
    for c in ctxs:
        c.synchronize()
    return [dict(state=c.opt_get(), vectors=c.opt_vectors(), rows=c.rows(), costs=c.costs()) for c in ctxs]


def assert_same(got, want, gens):
    assert len(got) == len(want)
    for g, (a, b) in enumerate(zip(got, want)):
        sa, sb = a["state"], b["state"]
        assert sa["itr"] == sb["itr"] == gens, (g, sa["itr"], sb["itr"])
        for k in ("sigma", "s", "best_f", "counteval", "iterator_sz"):
            assert sa[k] == sb[k], (g, k, sa[k], sb[k])
        np.testing.assert_array_equal(sa["xmean"], sb["xmean"], err_msg="shard %d" % g)
        for k in a["vectors"]:
            np.testing.assert_array_equal(a["vectors"][k], b["vectors"][k], err_msg="shard %d %s" % (g, k))
        np.testing.assert_array_equal(a["rows"], b["rows"], err_msg="shard %d rows" % g)
        np.testing.assert_array_equal(a["costs"], b["costs"], err_msg="shard %d costs" % g)
    for a in got[1:]:   # the replicas hold the same state
        np.testing.assert_array_equal(a["state"]["xmean"], got[0]["state"]["xmean"])
        assert a["state"]["sigma"] == got[0]["state"]["sigma"]


@pytest.mark.parametrize("shape,G,gens", [("cfg2", 2, 24), ("cfg2", 4, 24), ("cfg2_full_shard", 2, 20), ("cfg3", 4, 40),
                                          ("cfg1", 8, 12)])
def test_group_loop_gives_the_bits_of_the_phase_form(shape, G, gens):
    from lmcma_path_planner_amd import grids, hip
    cfg, nloc = SHAPES[shape]
    prob = grids.benchmark_problem(cfg)
    want = by_phases(prob, G, nloc, gens)
    ctxs = make_shards(prob, G, nloc)
    hip.HipContext.run_sharded_group(ctxs, 1)
    hip.HipContext.run_sharded_group(ctxs, gens // 2)
    hip.HipContext.run_sharded_group(ctxs, gens - 1 - gens // 2)
    got = snapshot(ctxs)
    for c in ctxs:
        c.close()
    assert_same(got, want, gens)


@pytest.mark.parametrize("shape,G,gens", [("cfg2", 2, 24), ("cfg2_full_shard", 2, 20), ("cfg3", 2, 40)])
def test_direct_exchange_in_one_process_gives_the_bits_of_the_phase_form(shape, G, gens):
    """Two contexts of this process, each on its own stream, connected through their mailboxes: each
    shard's call queues its whole share of the work; its kernels wait for the others' arrivals.
    (The kernels of all shards must be able to run side by side: a process has four hardware queues
    here, and streams that share one run in order, so more shards than that in ONE process on ONE
    GPU would wait for each other for ever; lmcma_hip_run_sharded_group is the form for that case,
    and one process per shard, below, the form the direct exchange is made for.)"""
    from lmcma_path_planner_amd import grids
    cfg, nloc = SHAPES[shape]
    prob = grids.benchmark_problem(cfg)
    want = by_phases(prob, G, nloc, gens)
    ctxs = make_shards(prob, G, nloc)
    handles = [c.peer_handle() for c in ctxs]
    for r, c in enumerate(ctxs):
        c.peer_connect(handles, r)
    for n in (1, gens // 2, gens - 1 - gens // 2):
        for c in ctxs:
            c.run_sharded(n)
    got = snapshot(ctxs)
    for c in ctxs:
        st = c.peer_status()
        assert st["wait_error"] == 0, st
    for c in ctxs:
        c.close()
    assert_same(got, want, gens)


@pytest.mark.parametrize("shape,G,gens", [("cfg2", 2, 24), ("cfg2", 4, 16), ("cfg2_full_shard", 8, 12)])
def test_direct_exchange_over_processes_gives_the_bits_of_the_phase_form(shape, G, gens, tmp_path):
    """One process per shard, all on the one GPU here; the mailboxes are mapped through IPC handles
    carried by torch.distributed (gloo).  Rank 0 compares with the phase form."""
    cfg, nloc = SHAPES[shape]
    port = 29650 + G
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "gpu_peer_worker.py"), str(cfg), str(nloc),
           str(gens)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "direct exchange over %d processes: identical" % G in out.stdout, out.stdout[-2000:]


def make_whole(prob, lam, count, seed=42):
    """`count` contexts that each hold the whole population of lam candidates."""
    from lmcma_path_planner_amd import hip
    ctxs = []
    for _ in range(count):
        c = hip.HipContext(prob.N, lam, prob.m, dims=prob.dims, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=seed,
                           w_len=prob.weights[0], w_grid=prob.weights[1], w_smooth=prob.weights[2])
        c.set_grid(prob.cells)
        c.set_bounds(prob.lo, prob.hi)
        c.opt_init(prob.x0, prob.sigma0)
        ctxs.append(c)
    return ctxs


ROWS_SHAPES = {  # name: (benchmark problem, candidates of the whole population)
    "cfg2": (2, 4096),
    "cfg3": (3, 8192),
    "cfg2_in_tiles": (2, 32768),      # above one counting tile: ranked in tiles, then merged
    "cfg1": (1, 128),
}


@pytest.mark.parametrize("shape,G,gens", [("cfg2", 2, 24), ("cfg2", 8, 24), ("cfg3", 4, 40), ("cfg2_in_tiles", 4, 20),
                                          ("cfg1", 8, 12)])
def test_rows_exchange_gives_the_bits_of_one_context(shape, G, gens):
    """The rows exchange (every context holds the whole population, samples its range, receives the
    other ranges' rows and costs, ranks and recombines everything itself) against ONE context running
    lmcma_hip_run: state, vectors, all rows and all costs bit for bit, whatever G.  Both ways of
    driving it: lmcma_hip_run_sharded_group, and phase by phase with the copies made here."""
    import torch
    from lmcma_path_planner_amd import dist as shard_mod
    from lmcma_path_planner_amd import grids, hip
    cfg, lam = ROWS_SHAPES[shape]
    prob = grids.benchmark_problem(cfg)
    one = make_whole(prob, lam, 1)
    one[0].run(gens)
    want = snapshot(one) * G
    one[0].close()

    ctxs = make_whole(prob, lam, G)
    hip.HipContext.run_sharded_group(ctxs, 1)
    hip.HipContext.run_sharded_group(ctxs, gens // 2)
    hip.HipContext.run_sharded_group(ctxs, gens - 1 - gens // 2)
    got = snapshot(ctxs)
    for c in ctxs:
        c.close()
    assert_same(got, want, gens)

    dev = torch.device("cuda", 0)
    ctxs = make_whole(prob, lam, G)
    cnt, ld = lam // G, ctxs[0].row_stride()
    rows = [shard_mod.alias_tensor(c, hip.BUF_ROWS_LOCAL, torch.float32, dev) for c in ctxs]
    cost = [shard_mod.alias_tensor(c, hip.BUF_COST_ALL, torch.float32, dev) for c in ctxs]
    for _ in range(gens):
        for r, c in enumerate(ctxs):
            c.phase_local_range(r * cnt, cnt)
        for c in ctxs:
            c.synchronize()
        for r in range(G):
            for p in range(G):
                if p != r:
                    rows[r][p * cnt * ld:(p + 1) * cnt * ld].copy_(rows[p][p * cnt * ld:(p + 1) * cnt * ld])
                    cost[r][p * cnt:(p + 1) * cnt].copy_(cost[p][p * cnt:(p + 1) * cnt])
        torch.cuda.synchronize()
        for c in ctxs:
            c.phase_rank()
            c.phase_update(0)
    got = snapshot(ctxs)
    for c in ctxs:
        c.close()
    assert_same(got, want, gens)


def test_rows_exchange_errors():
    from lmcma_path_planner_amd import _lib, grids, hip
    prob = grids.benchmark_problem(1)
    a, b = make_whole(prob, 128, 2)
    with pytest.raises(_lib.LmcmaError):
        a.phase_local_range(0, 130)                 # beyond the population
    with pytest.raises(_lib.LmcmaError):
        a.phase_local_range(2, 64)                  # not on a multiple of 4
    shard = make_shards(prob, 2, 64)
    with pytest.raises(_lib.LmcmaError):
        shard[0].phase_local_range(0, 64)           # a shard that exchanges sorted lists
    with pytest.raises(_lib.LmcmaError):
        hip.HipContext.run_sharded_group([a, shard[1]], 1)      # a whole population and a shard
    c3 = make_whole(prob, 128, 1)[0]
    with pytest.raises(_lib.LmcmaError):
        hip.HipContext.run_sharded_group([a, b, c3], 1)         # 128 candidates do not divide by 3
    for c in (a, b, c3, *shard):
        c.close()


def test_a_peer_that_never_arrives_is_reported_not_waited_for_forever():
    from lmcma_path_planner_amd import _lib, grids
    prob = grids.benchmark_problem(1)
    ctxs = make_shards(prob, 2, 64)
    handles = [c.peer_handle() for c in ctxs]
    for r, c in enumerate(ctxs):
        c.peer_connect(handles, r)
    ctxs[0].run_sharded(1)          # shard 1 never runs: the first wait of shard 0 gives up after 4 s
    ctxs[0].synchronize()
    assert ctxs[0].peer_status()["wait_error"] == 2     # 1 + the source that did not arrive
    with pytest.raises(_lib.LmcmaError):
        ctxs[0].opt_get()
    for c in ctxs:
        c.close()


def test_peer_errors():
    from lmcma_path_planner_amd import _lib, grids
    prob = grids.benchmark_problem(1)
    a, b = make_shards(prob, 2, 64)
    with pytest.raises(_lib.LmcmaError):
        a.peer_connect([b"\0" * 128, b"\0" * 128], 0)       # no mailbox yet
    ha, hb = a.peer_handle(), b.peer_handle()
    with pytest.raises(_lib.LmcmaError):
        a.peer_connect([ha, b"\0" * 128], 0)                # not a handle
    with pytest.raises(_lib.LmcmaError):
        a.peer_connect([hb, ha], 0)                         # own handle in the wrong place
    with pytest.raises(_lib.LmcmaError):
        a.peer_connect([ha], 0)                             # the context is half of its population
    a.peer_connect([ha, hb], 0)
    with pytest.raises(_lib.LmcmaError):
        a.peer_connect([ha, hb], 0)                         # connected already
    a.peer_disconnect()
    with pytest.raises(_lib.LmcmaError):
        a.run_sharded(1)                                    # neither communicator nor peers
    a.close()
    b.close()


def test_bench_starts_its_own_ranks(tmp_path):
    """bench.py --gpus N without a launcher around it: it starts N ranks itself (here sharing the one
    GPU: --oversubscribe) and rank 0 prints the one JSON line with n_gpus = N; without the switch it
    refuses to measure fewer GPUs than it was asked for."""
    import json
    import torch
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "30",
                          "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["lambda_total"] == 8192
    assert rec["peer_status"]["wait_error"] == 0 and rec["value"] > 0
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"],
                             capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert out.returncode != 0 and "refusing" in (out.stdout + out.stderr)
