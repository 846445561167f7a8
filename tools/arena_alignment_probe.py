"""Arena-internal alignment of the per-(batch, column) arrays (mi_hbm_options.array_align) vs decode time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import duckdb_arrow_amd as da
from duckdb_arrow_amd import hbm
buf, info = da.synth_lineitem_stream(scale_factor=10.0, seed=42)
ctx = da.Context(0)
for align in (256, 4096, 65536, 2 << 20):
    hs = hbm.HbmStream(ctx, buf, array_align=align)
    for _ in range(30):
        hs.launch()
    hs.status()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            hs.launch()
        hs.status()
        ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print("array alignment %8d  %.3f ms/step" % (align, float(np.median(ts))), flush=True)
    hs.close()
