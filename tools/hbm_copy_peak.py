"""Measured device-memory copy rate of this GPU (read + write bytes per second), to quote beside the
nominal 8 TB/s the roofline fraction divides by.   python tools/hbm_copy_peak.py"""
import torch

dev = torch.device("cuda", 0)
for gib in (0.25, 1, 4):
    n = int(gib * (1 << 30)) // 4
    x = torch.ones(n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("copy of %.2f GiB: %.3f ms, %.2f TB/s (read + write)" % (gib, ms, 2 * n * 4 / ms / 1e9))
    z = torch.empty(1, device=dev)
    e0.record()
    for _ in range(reps):
        z = x.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("read (sum) of %.2f GiB: %.3f ms, %.2f TB/s" % (gib, ms, n * 4 / ms / 1e9))
    del x, y
