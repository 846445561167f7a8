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


def _worker_file_list(rank, world, port, paths, out_q):
    """BASELINE config 3 on CPU: every rank walks the file list with the product's host reader (mi_reader_index), takes the
    record batches whose global ordinal k (files first, then batches inside a file) has k mod world == rank -- the rule
    mi_scan_options.rank / world implements on the GPU path -- decodes them with the oracle and applies the l_shipdate
    range with the oracle's filter."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import duckdb_arrow_amd as da
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ordinal, mine, rows, selected = 0, [], 0, 0
    for path in paths:
        data = np.fromfile(path, np.uint8)
        rd = da.Reader(path=path)
        rd.schema()
        idx = [e for e in rd.index() if e["type"] == 3]   # RecordBatch messages
        rd.close()
        head = data[: idx[0]["prefix_offset"]] if idx else data
        for e in idx:
            if ordinal % world == rank:
                mine.append(ordinal)
                end = e["body_offset"] + e["body_len"]
                sub = np.concatenate([head, data[e["prefix_offset"]: end], np.frombuffer(b"\xff\xff\xff\xff\x00\x00\x00\x00", np.uint8)])
                _, dec = po.decode_stream(sub)
                cols = {c["name"]: c for c in dec[0]["columns"]}
                n = dec[0]["nrows"]
                ship = cols["l_shipdate"]["data"].view(np.int32)
                sel = po.filter_cnf([[("l_shipdate", ">=", 8766)], [("l_shipdate", "<", 9131)]], {"l_shipdate": (ship, cols["l_shipdate"]["validity"])}, n)
                rows += n
                selected += len(sel)
            ordinal += 1
    t = torch.tensor([rows, selected, ordinal], dtype=torch.int64)
    dist.barrier()
    dist.all_reduce(t[:2], op=dist.ReduceOp.SUM)
    out_q.put((rank, mine, int(t[0]), int(t[1]), ordinal))
    dist.destroy_process_group()


def test_two_ranks_share_a_file_list_with_the_shipdate_filter(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import duckdb_arrow_amd as da
    # one 90000-row table as 8 files of 1-2 record batches each
    paths, total, want = [], 0, 0
    rpb = 8000
    whole, info = da.synth_lineitem_stream(scale_factor=1.0, seed=11, n_rows=90000, rows_per_batch=rpb, n_threads=2)
    offs, nb = info["batch_offsets"], info["n_batches"]
    per = (nb + 7) // 8
    for i in range(8):
        lo, hi = offs[min(nb, i * per)], offs[min(nb, (i + 1) * per)]
        p = str(tmp_path / ("part_%d.arrows" % i))
        with open(p, "wb") as f:
            f.write(whole[: offs[0]].tobytes() + whole[lo:hi].tobytes() + b"\xff\xff\xff\xff\x00\x00\x00\x00")
        paths.append(p)
    from oracle import pyoracle as po
    _, dec = po.decode_stream(whole)
    for b in dec:
        ship = {c["name"]: c for c in b["columns"]}["l_shipdate"]["data"].view(np.int32)
        want += int(((ship >= 8766) & (ship < 9131)).sum())
        total += b["nrows"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file_list, args=(r, 2, port, paths, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][4] == res[1][4] == nb == 12
    assert sorted(res[0][1] + res[1][1]) == list(range(nb)) and all(k % 2 == 0 for k in res[0][1]) and all(k % 2 == 1 for k in res[1][1])
    assert res[0][2] == res[1][2] == total == 90000 and res[0][3] == res[1][3] == want
