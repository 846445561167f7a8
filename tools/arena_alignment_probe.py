"""Arena-internal alignment of the per-(batch, column) arrays vs decode time, at several arena offsets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import duckdb_arrow_amd as da
from duckdb_arrow_amd import hbm
buf, info = da.synth_lineitem_stream(scale_factor=10.0, seed=42)
ctx = da.Context(0)
stream = torch.cuda.current_stream().cuda_stream
orig_zeros = torch.zeros
orig_round = hbm._round_up
for align in (256, 4096, 65536, 2 << 20):
    hbm._round_up = lambda v, a=256, _al=align: orig_round(v, max(a, _al))
    res = []
    for off in [0, 2048, 4096, 1 << 20, 16 << 20, 32 << 20]:
        big = {}
        def zeros(n, dtype=None, device=None):
            if isinstance(n, int) and n > (1 << 30):
                big["t"] = orig_zeros(n + (128 << 20), dtype=dtype, device=device)
                return big["t"][off: off + n]
            return orig_zeros(n, dtype=dtype, device=device)
        torch.zeros = zeros
        hs = hbm.HbmStream(ctx, buf)
        torch.zeros = orig_zeros
        for _ in range(30):
            hs.launch(stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                hs.launch(stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        res.append("%d:%.3f" % (off, float(np.median(ts))))
        del hs, big
        torch.cuda.empty_cache()
    print("array alignment %8d  " % align + "  ".join(res), flush=True)
hbm._round_up = orig_round
