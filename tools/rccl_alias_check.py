import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
from lmcma_path_planner_amd import grids, hip, dist as shard_mod
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
device = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
prob = grids.benchmark_problem(2)
st = torch.cuda.Stream(device); torch.cuda.set_stream(st)
ctx = hip.HipContext(prob.N, prob.lam, prob.m, dims=2, waypoints=prob.W, start=prob.start, goal=prob.goal, seed=1, stream=st.cuda_stream)
ctx.set_grid(prob.cells); ctx.set_bounds(prob.lo, prob.hi); ctx.opt_init(prob.x0, prob.sigma0)
sh = shard_mod.HipShard(ctx, device)
for g in range(5):
    sh.phase_local()
    before = sh.cost_local.clone()
    dist.all_gather_into_tensor(sh.cost_all, sh.cost_local)          # in place, library memory
    sh.phase_rank()
    L = sh.partial_len
    dist.all_gather_into_tensor(sh.partial_all[:L], sh.partial_local)
    sh.phase_update(1)
    torch.cuda.synchronize()
    assert torch.equal(before, sh.cost_local)
print("rccl on aliased library buffers: ok, itr", ctx.opt_get()["itr"], "sigma", ctx.opt_get()["sigma"])
ctx.close(); dist.destroy_process_group()
