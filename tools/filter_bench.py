"""Kernel-level timing of the K6 range filter on 240 M resident int32 values (wall clock around 20 launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import duckdb_arrow_amd as da
ctx = da.Context(0)
n = 240_000_000
vals = torch.randint(8036, 10562, (n,), dtype=torch.int32, device="cuda")
sel = torch.empty(n, dtype=torch.int32, device="cuda")
cnt = torch.zeros((n + 2047) // 2048, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    da.filter_range(ctx, vals.data_ptr(), 4, 0, n, 8766, 9131, sel.data_ptr(), cnt.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    da.filter_range(ctx, vals.data_ptr(), 4, 0, n, 8766, 9131, sel.data_ptr(), cnt.data_ptr(), s)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
k = int(cnt.sum().item())
print("filter_range int32: %.3f ms, %.1f G rows/s, selected %.4f, alg bytes %.2f GB -> %.0f GB/s" % (ms, n / ms / 1e6, k / n, (4 * n + 4 * k + cnt.numel() * 4) / 1e9, (4 * n + 4 * k + cnt.numel() * 4) / ms / 1e6))
