"""Time of the one-workgroup sort of a 4096-key tile (k_sort_tiles through lmcma_hip_rank_noalloc).
    python tools/sort_tile_time.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from lmcma_path_planner_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
for lam in (1024, 4096, 8192):
    cost = torch.rand(lam, device=dev, dtype=torch.float32)
    sc = torch.empty(lam, device=dev, dtype=torch.float32)
    perm = torch.empty(lam, device=dev, dtype=torch.int32)
    for _ in range(5):
        assert lib.lmcma_hip_rank_noalloc(cost.data_ptr(), lam, sc.data_ptr(), perm.data_ptr(), st.cuda_stream) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record(st)
    for _ in range(reps):
        lib.lmcma_hip_rank_noalloc(cost.data_ptr(), lam, sc.data_ptr(), perm.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ref = torch.sort(cost, stable=True)
    assert torch.equal(ref.values, sc) and torch.equal(ref.indices.int(), perm)
    print("lambda %d: %.2f us per call" % (lam, e0.elapsed_time(e1) * 1000 / reps))
