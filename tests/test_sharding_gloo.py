"""The N>1 path on CPU: two ranks over gloo, each generating and indexing its own row-group shard exactly like
bench.py does (no data-path collective; the only communication is the barrier + the reductions of the timing
harness).  Host-only: generator, IPC reader and the CPU oracle -- no GPU needed."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows_per_rank, rows_per_batch, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import duckdb_arrow_amd as da
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batches = (rows_per_rank + rows_per_batch - 1) // rows_per_batch
    buf, info = da.synth_lineitem_stream(scale_factor=1.0, seed=5, n_rows=rows_per_rank, rows_per_batch=rows_per_batch,
                                         first_row=rank * batches * rows_per_batch, n_threads=2)
    idx = da.Reader(buffers=[buf]).index()
    rc, st = po.scan_stream(buf, want_checksum=True)
    assert rc == 0
    rows = torch.tensor([st["rows"]], dtype=torch.int64)
    dist.barrier()
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)                 # whole-job row count
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # max-over-ranks time, as bench.py reports it
    out_q.put((rank, int(rows.item()), float(t.item()), len(idx), st["checksum"], buf[idx[0]["body_offset"]: idx[0]["body_offset"] + 4096].tobytes()))
    dist.destroy_process_group()


def test_two_ranks_shard_row_groups_without_a_collective():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    rows_per_rank, rpb = 30000, 10000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rows_per_rank, rpb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [60000, 60000] and [r[2] for r in res] == [1.5, 1.5] and [r[3] for r in res] == [3, 3]
    # the two shards are the two halves of ONE 60000-row table: rank 1's first batch is global batch 3
    sys.path.insert(0, ROOT)
    import duckdb_arrow_amd as da
    whole, _ = da.synth_lineitem_stream(scale_factor=1.0, seed=5, n_rows=60000, rows_per_batch=rpb, n_threads=2)
    widx = da.Reader(buffers=[whole]).index()
    assert len(widx) == 6
    assert res[0][5] == whole[widx[0]["body_offset"]: widx[0]["body_offset"] + 4096].tobytes()
    assert res[1][5] == whole[widx[3]["body_offset"]: widx[3]["body_offset"] + 4096].tobytes()
    assert res[0][4] != res[1][4]
