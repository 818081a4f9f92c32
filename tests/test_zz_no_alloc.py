"""lmcma_hip_rank_noalloc (the ranking for callers that cannot allocate) against lmcma_hip_select,
through the C ABI: two calls in a row, the second on a smaller population."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _costs(rng, lam):
    costs = rng.gamma(2.0, 500.0, size=lam)
    costs[rng.integers(0, lam, size=lam // 4)] = 123.25           # many ties
    costs[3] = np.inf
    costs[5] = np.nan
    costs[7] = -0.0
    costs[9] = 0.0
    costs[11] = -5.0
    return costs


def test_rank_noalloc_equals_select_on_two_calls_in_a_row():
    import torch
    from lmcma_path_planner_amd import _lib, hip
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    rng = np.random.default_rng(12)
    # buffers sized for the larger call, allocated before either call; the second call leaves
    # their tails alone
    big, small = 50000, 3000
    d_cost = torch.empty(big, dtype=torch.float32, device=dev)
    d_sorted = torch.full((big,), -1.0, dtype=torch.float32, device=dev)
    d_perm = torch.full((big,), -1, dtype=torch.int32, device=dev)
    for lam in (big, small):
        costs = _costs(rng, lam)
        ctx = hip.HipContext(4, lam, 1, z_source=hip.Z_HOST, cost_source=hip.COST_EXTERNAL)
        ctx.set_costs(costs)
        want_sorted, want_perm = ctx.select()
        ctx.close()
        d_cost[:lam].copy_(torch.from_numpy(costs.astype(np.float32)))
        d_sorted.fill_(-1.0)
        d_perm.fill_(-1)
        torch.cuda.synchronize()
        rc = lib.lmcma_hip_rank_noalloc(ctypes.c_void_p(d_cost.data_ptr()), lam, ctypes.c_void_p(d_sorted.data_ptr()),
                                        ctypes.c_void_p(d_perm.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, lib.lmcma_hip_last_error()
        stream.synchronize()
        np.testing.assert_array_equal(d_perm[:lam].cpu().numpy(), want_perm)           # bit-exact: integer work
        np.testing.assert_array_equal(d_sorted[:lam].cpu().numpy(), want_sorted)
        assert torch.all(d_perm[lam:] == -1) and torch.all(d_sorted[lam:] == -1.0)     # nothing written past lambda
    # sizes beyond the fixed scratch array are refused, with a message
    rc = lib.lmcma_hip_rank_noalloc(ctypes.c_void_p(d_cost.data_ptr()), 131073, ctypes.c_void_p(d_sorted.data_ptr()),
                                    ctypes.c_void_p(d_perm.data_ptr()), None)
    assert rc == -4 and b"131072" in lib.lmcma_hip_last_error()
    rc = lib.lmcma_hip_rank_noalloc(None, 16, ctypes.c_void_p(d_sorted.data_ptr()), ctypes.c_void_p(d_perm.data_ptr()), None)
    assert rc == -1
