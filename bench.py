#!/usr/bin/env python3
"""bench.py -- TPC-H lineitem.arrows full-column scan on MI355X: rows/s + achieved HBM GB/s (BASELINE.json metric).

One STEP = one pass of the hot path over the whole resident table: every column of every record batch of a
synthetic `lineitem.arrows` (122880-row batches, DuckDB export schema, validity bitmaps present) is transcoded
from Arrow IPC buffers to DuckDB vectors by the HIP kernels (3 launches per step: copy, dec128, string).  The IPC
stream is resident in HBM when the timed region starts (PCIe-inclusive numbers ride along as `operator_path`, never
`value`).  Everything goes through the C ABI (mi_hbm_* / mi_scan_*); torch only provides the stream handle, the
barrier and the max-over-ranks reduction.

  N = 1   workload = BASELINE.json configs[1]: TPC-H SF10 lineitem (59 986 052 rows, 489 record batches) on one GPU.
  N > 1   one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks reduction):
          record batches shard embarrassingly, so every rank scans its own SF10-sized shard of an SF(10*N) table
          (weak scaling, no data-path collective); value = rows of all ranks / max-over-ranks time.
          Launch contract: under a launcher (torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK / MASTER_*) the process
          is one rank.  WITHOUT one, `python bench.py --gpus N` starts the N ranks itself: the parent -- which never touches
          the GPU and never imports torch -- spawns N fresh child processes with those variables set, relays rank 0's JSON
          line and exits non-zero if any child does.  `n_gpus` is the number of ranks that ran.

Besides the contract's fields the JSON line carries
  roofline           HBM roofline of the dominant kernel, from HIP-event timings taken live (mi_hbm_launch_timed)
  kernels            the same for every kernel class
  reference_shaped   secondary, never `value`: the same scan laid out the way the reference's vectors are (plain
                     fixed-width columns alias the Arrow buffer, all-valid columns carry no mask) with its own roofline
  operator_path      secondary: the scan OPERATOR (file in /dev/shm -> pread -> pinned -> H2D -> kernels -> D2H) over
                     the SF10 table written as 8 files, every rank taking its share of the record batches
                     (rank / world): full-column host-consumer scan and BASELINE config 3's l_shipdate pushdown
  cpu_baseline       the CPU oracle (a port of the reference's scan path, single thread like the reference's
                     single-file scan) timed on a bounded sample of the same stream, rank 0 / N=1 only
  cpu_baseline_encode  the same for the COPY TO direction (K7), beside BASELINE config 4's kernel numbers
  parity             sampled record batches of the measured run compared bit for bit with the oracle
  sf100              secondary, N = 1 only: the configuration north_star states its target on (TPC-H SF100, one GPU), run in a
                     child process after the SF10 legs when HBM and host memory allow
  north_star_read_roofline   the north star's own yardstick (bytes READ per second / 8 TB/s, target 0.6) and why it is not met
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 TB/s achievable copy)
SHIP_LO, SHIP_HI = 8766, 9131   # config 3: 1994-01-01 <= l_shipdate < 1995-01-01


def kernel_table(cstats, per_class, pmc, same_workload):
    kernels = []
    for cs, ms in zip(cstats, per_class):
        if cs["tiles"] == 0:
            continue
        b = cs["bytes_read"] + cs["bytes_written"]
        kernels.append({"kernel": cs["kernel"], "ms": float(ms), "algorithmic_bytes": b,
                        "achieved_GBps": b / (ms * 1e-3) / 1e9 if ms > 0 else None,
                        "frac_of_8TBps": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
                        "traffic": pmc.get(cs["kernel"]) if same_workload else None})
    return kernels


def roofline_of(kernels, alg_bytes, ms_per_step, traffic_stale=False):
    dom = max(kernels, key=lambda k: k["ms"])
    return {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac_of_8TBps"], "traffic": dom["traffic"],
            "traffic_stale": bool(traffic_stale) if dom["traffic"] else None,   # the kernel sources changed after the PMC passes
            "traffic_source": "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                              "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch)" if dom["traffic"] else None,
            "algorithmic_bytes": dom["algorithmic_bytes"],
            "whole_step_frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}


def read_roofline(rows_per_s_per_gpu, read_bytes_per_row, written_bytes_per_row):
    """north_star's yardstick: Arrow bytes READ per second against the 8 TB/s peak, target 0.6."""
    frac = rows_per_s_per_gpu * read_bytes_per_row / 1e9 / HBM_PEAK_GBS
    need = 0.6 * HBM_PEAK_GBS * (read_bytes_per_row + written_bytes_per_row) / read_bytes_per_row
    return {"frac": frac, "target": 0.6, "met": bool(frac >= 0.6), "read_GBps": rows_per_s_per_gpu * read_bytes_per_row / 1e9,
            "why": None if frac >= 0.6 else
            "every row read (%.1f B) is also written as DuckDB vectors (%.1f B): 60 %% of the READ roofline would need %.0f GB/s of total "
            "HBM traffic, %.2fx the chip's %.0f GB/s peak; the whole step runs at `roofline.whole_step_frac` of that peak"
            % (read_bytes_per_row, written_bytes_per_row, need, need / HBM_PEAK_GBS, HBM_PEAK_GBS)}


def cpu_multifile_baseline(paths):
    """oracle_scan.c (body copy + FULL offset validation + 2048-row pull loop) over the files of the operator-path legs,
    one file per thread; the files are read into memory first (untimed: they sit in the page cache, and the scan's body copy
    stands for the reference's ReadData).  ctypes releases the GIL around the C call."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as po
    bufs = [np.fromfile(p, dtype=np.uint8) for p in paths]
    n_threads = max(1, min(len(bufs), os.cpu_count() or 1))

    def one(b):
        rc, st = po.scan_stream(b)
        assert rc == 0
        return st["rows"]

    dt = None
    with ThreadPoolExecutor(n_threads) as ex:
        for _ in range(2):   # the first pass also touches every thread's scratch for the first time
            t1 = time.perf_counter()
            rows = sum(ex.map(one, bufs))
            d1 = time.perf_counter() - t1
            dt = d1 if dt is None else min(dt, d1)
    return {"value": rows / dt, "unit": "rows/s", "cores": n_threads, "kind": "port", "seconds": dt, "rows": rows,
            "sample": "the whole table: %d files, one oracle_scan.c scan per thread" % len(bufs), "host_cpus": os.cpu_count()}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one rank each.  The parent makes no GPU call
    (it does not even import torch); rank 0's stdout is the result line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_id = "%d_%d" % (os.getpid(), port)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MI_BENCH_RUN_ID=run_id, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=120 if codes[0] == 0 else 5))
        except subprocess.TimeoutExpired:   # a rank that outlives rank 0: end exactly that process
            p.kill()
            codes.append(p.wait())
    line = [ln for ln in (out0 or "").strip().split("\n") if ln.startswith("{")]
    if line:
        print(line[-1])
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or not line:
        print("bench.py: ranks failed (rank, exit code): %s%s" % (bad, "" if line else "; rank 0 printed no result line"), file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def run_sf100_child(args, torch):
    """The configuration north_star states its target on -- TPC-H SF100 lineitem on ONE GPU (105 GB of Arrow buffers + 95 GB
    of vectors resident in the 288 GB of HBM) -- as a child process of its own after the SF10 legs: fresh HBM, and the
    105 GB host copy of the stream goes away with it."""
    import subprocess
    free_hbm, _ = torch.cuda.mem_get_info()
    avail_host = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail_host = int(ln.split()[1]) * 1024
    except OSError:
        pass
    if free_hbm < 210e9 or avail_host < 150e9:
        return {"skipped": "needs 210 GB of free HBM and 150 GB of host memory; free: %.0f GB HBM, %.0f GB host" % (free_hbm / 1e9, avail_host / 1e9)}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sf", "100", "--steps", "5", "--warmup", "2", "--no-operator-path",
           "--no-cpu-baseline", "--no-encode-leg", "--no-sf100", "--seed", str(args.seed)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after 420 s"}
    line = [ln for ln in run.stdout.strip().split("\n") if ln.startswith("{")]
    if run.returncode != 0 or not line:
        return {"error": (run.stderr or "")[-300:]}
    z = json.loads(line[-1])
    rs = z.get("reference_shaped") or {}
    return {"workload": z["config"]["workload"], "steps": z["steps"], "warmup": z["warmup"],
            "ms_per_step": z["ms_per_step"], "rows_per_s": z["value"], "achieved_hbm_GBps_whole_step": z["achieved_hbm_GBps_whole_step"],
            "roofline": z["roofline"], "read_roofline_frac": z["north_star_read_roofline"]["frac"], "parity": z["parity"],
            "reference_shaped": {k: rs.get(k) for k in ("ms_per_step", "rows_per_s", "achieved_GBps", "roofline", "read_roofline_frac")},
            "setup_seconds": z["setup_seconds"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=10.0, help="TPC-H scale factor per GPU (default 10 = configs[1])")
    ap.add_argument("--rows", type=int, default=0, help="override the row count per GPU (smoke runs)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-validity", action="store_true", help="pyarrow-style stream without validity bitmaps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-operator-path", action="store_true", help="skip the /dev/shm operator-path legs")
    ap.add_argument("--no-numa-bind", action="store_true", help="leave this process's threads where the scheduler puts them (default: the GPU's NUMA node)")
    ap.add_argument("--no-encode-leg", action="store_true", help="skip the K7 encode kernels over the decoded vectors (config 4)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (decode; the encode leg gets half)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="untimed clock ramp before the warmup steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-process path with all ranks on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--shm-dir", default="/dev/shm", help="where the operator-path legs put their files")
    ap.add_argument("--no-sf100", action="store_true", help="skip the SF100 block (N = 1; a child process after the SF10 legs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])   # never returns; nothing above has touched the GPU or imported torch

    # the library first: loading it settles the HIP runtime's hardware-queue count (c_api.cpp MiRuntimeDefaults) before
    # anything -- torch included -- makes the process's first HIP call
    import duckdb_arrow_amd as da
    da._ffi.lib()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(local_rank)
    n_gpus = world   # the ranks that run
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but the launcher started WORLD_SIZE=%d ranks; n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)

    from duckdb_arrow_amd.hbm import HbmStream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    # ---- this rank's shard: SF(sf) rows starting at rank * rows (row groups of one SF(sf*N) table) ----
    rows_per_batch = 122880
    n_rows = args.rows if args.rows else {1.0: 6001215, 10.0: 59986052, 100.0: 600037902}.get(args.sf, int(6001215 * args.sf))
    batches_per_rank = (n_rows + rows_per_batch - 1) // rows_per_batch
    first_row = rank * batches_per_rank * rows_per_batch
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=args.seed, n_rows=n_rows, rows_per_batch=rows_per_batch,
                                         with_validity=not args.no_validity, n_threads=threads, first_row=first_row)
    t_gen = time.time() - t0

    ctx = da.Context(local_rank)
    # the host side of this rank on its GPU's NUMA node (a deployment pins a worker the same way): the page cache of the files
    # the operator-path legs write, the consumer thread, the CPU baselines' threads; the library's own threads go there by
    # themselves (MI_NUMA_BIND).  Reported in the line.
    numa_node = -1 if args.no_numa_bind else ctx.bind_this_thread()
    t0 = time.time()
    hs = HbmStream(ctx, buf, device="cuda:%d" % local_rank)   # mi_hbm_open: parse + upload + layout + plan, all in the library
    torch.cuda.synchronize()
    t_upload = time.time() - t0
    stats = hs.stats()
    cstats = hs.plan.class_stats()
    stream = torch.cuda.current_stream().cuda_stream

    def timed_steps(h):
        """W untimed warmup steps, then exactly K steps between barriers; returns the max-over-ranks seconds."""
        for _ in range(args.warmup):
            h.launch(stream)
        torch.cuda.synchronize()
        assert h.status() == 0, "device status after warmup"
        barrier()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            h.launch(stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_start
        barrier()
        assert h.status() == 0, "device status"
        return max_over_ranks(dt)

    # ---- setup: bring the GPU out of its idle clock state (a fresh process measures ~4 % slower for the first
    # few hundred milliseconds) ----
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        for _ in range(10):
            hs.launch(stream)
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps of the fully materialised scan (`value`) ----
    elapsed = timed_steps(hs)

    # ---- per-kernel HIP-event timings (same stream, same launches, outside the timed region) ----
    def per_class_ms(h):
        acc = np.zeros(len(cstats))
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            acc += np.array(h.plan.launch_timed(stream))
        return acc / reps

    per_class = per_class_ms(hs)
    traffic_src = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    pmc_doc = json.load(open(traffic_src)) if os.path.exists(traffic_src) else {}
    pmc = pmc_doc.get("traffic_bytes_per_launch", {})
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_sha import kernel_source_sha
    traffic_stale = bool(pmc) and pmc_doc.get("kernel_source_sha") != kernel_source_sha()
    same_workload = args.sf == 10.0 and not args.rows and not args.no_validity

    # ---- secondary, never `value`: the reference-shaped layout over the same resident stream -- plain fixed-width columns
    # alias the Arrow buffer in HBM (DirectConversion) and all-valid columns carry no validity words (unset ValidityMask) ----
    ref_shaped = None
    if world == 1:
        zs = HbmStream(ctx, buf, zero_copy_direct=True, unset_all_valid=True, share_stream_of=hs)
        dz = timed_steps(zs)
        zst, zcs = zs.stats(), zs.plan.class_stats()
        zk = kernel_table(zcs, per_class_ms(zs), {}, False)
        z_alg = zst["bytes_read"] + zst["bytes_written"]
        aliased = sorted({e["name"] for lay in zs.layout[:1] for e in lay["columns"] if e["alias_off"] >= 0})
        unmasked = sorted({e["name"] for lay in zs.layout[:1] for e in lay["columns"] if e["valid_off"] < 0 and e["alias_off"] < 0})
        ref_shaped = {"ms_per_step": dz / args.steps * 1e3, "rows_per_s": info["n_rows"] * args.steps / dz,
                      "algorithmic_bytes_per_row": {"read": zst["bytes_read"] / info["n_rows"], "written": zst["bytes_written"] / info["n_rows"]},
                      "achieved_GBps": z_alg * args.steps / dz / 1e9,
                      "read_roofline_frac": info["n_rows"] * args.steps / dz * (stats["bytes_read"] / info["n_rows"]) / 1e9 / HBM_PEAK_GBS,
                      "roofline": roofline_of(zk, z_alg, dz / args.steps * 1e3), "kernels": zk,
                      "tasks": zs.n_tasks, "of_tasks": hs.n_tasks, "aliased_columns": aliased, "columns_without_validity_words": unmasked,
                      "note": "secondary figure, not `value`: mi_hbm_options.zero_copy_direct + unset_all_valid = the shape of the "
                              "reference's own vectors (DirectConversion points the vector at the Arrow buffer, an array without "
                              "NULLs leaves the ValidityMask unset); what a device-resident consumer gets by default"}
        zs.close()

    # ---- parity of the measured run: sampled batches of EVERY rank's shard vs the oracle ----
    parity = cpu_baseline = cpu_baseline_encode = None
    from oracle import pyoracle as po
    msgs = [m for m in po.walk_stream(buf) if m["type"] == po.MSG_RECORD_BATCH]
    sample = sorted(set([0, len(msgs) // 2, len(msgs) - 1]))
    got = hs.fetch(batches=sample)
    ok = True
    for bi in sample:
        m = msgs[bi]
        sub = np.concatenate([buf[: msgs[0]["prefix_off"]], buf[m["prefix_off"]: m["body_off"] + m["body_len"]]])
        shift = m["prefix_off"] - msgs[0]["prefix_off"]
        _, want = po.decode_stream(sub, ptr_base_of=lambda i, body_off, boff: body_off + boff + shift)
        for gc, wc in zip(got[bi]["columns"], want[0]["columns"]):
            ok = ok and np.array_equal(gc["data"], wc["data"]) and np.array_equal(gc["validity"], wc["validity"])
    del got
    ranks_ok = 1 if ok else 0
    if world > 1:
        t = torch.tensor([ranks_ok], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        ranks_ok = int(t.item())
    parity = {"checked_batches": sample, "bit_exact": bool(ranks_ok == world), "ranks_bit_exact": ranks_ok, "ranks": world}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            # bounded sample: time 2 batches, then as many as fit the budget
            rc, st = po.scan_stream(buf, max_batches=2)
            t1 = time.perf_counter()
            rc, st = po.scan_stream(buf, max_batches=2)
            per_batch = (time.perf_counter() - t1) / max(1, st["batches"])
            nb = int(max(2, min(len(msgs), args.cpu_seconds / max(per_batch, 1e-6))))
            # repeat the pass until ~cpu_seconds of CPU work are on the clock (SF10 is ~1.3 s per pass on one core)
            dt, rows, passes = 0.0, 0, 0
            while passes == 0 or (dt < args.cpu_seconds * 0.85 and passes < 64):
                t1 = time.perf_counter()
                rc, st = po.scan_stream(buf, max_batches=nb)
                dt += time.perf_counter() - t1
                assert rc == 0
                rows += st["rows"]
                passes += 1
            cpu_baseline = {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
                            "sample": "%d pass(es) over the first %d record batches (%d rows, %.2f GB of Arrow buffers) of the "
                                      "same stream, oracle_scan.c: body copy + FULL offset validation + 2048-row pull loop"
                                      % (passes, st["batches"], st["rows"], st["bytes_in"] / 1e9),
                            "seconds": dt, "host_cpus": os.cpu_count()}
            # the COPY TO direction (BASELINE config 4): K7 + the reference's two extra copies per record batch, one core
            rc, st = po.encode_stream(buf, max_batches=1)
            per_batch = max(st["seconds"], 1e-6)
            nb = int(max(1, min(len(msgs), args.cpu_seconds * 0.5 / per_batch)))
            rc, st = po.encode_stream(buf, max_batches=nb, verify=True)
            assert rc == 0 and st["mismatches"] == 0
            cpu_baseline_encode = {"value": st["rows"] / st["seconds"], "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": "the first %d record batches (%d rows) of the same table as DuckDB vectors, oracle_scan.c "
                                             "orc_encode_stream: chunk concatenation + ArrowAppender loops + body copy; every produced "
                                             "buffer equals the source stream's" % (st["batches"], st["rows"]),
                                   "seconds": st["seconds"], "GBps_out": st["bytes_out"] / st["seconds"] / 1e9}
    # ---- secondary, never `value`: BASELINE config 4 at kernel level -- the K7 encode kernels over the vectors this run has
    # just decoded (DuckDB vectors resident in HBM -> Arrow buffers), one plan for the table, HIP events per kernel class ----
    encode_kernels = None
    if rank == 0 and world == 1 and not args.no_encode_leg:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from encode_bench import encode_leg
        from duckdb_arrow_amd import _ffi
        e = encode_leg(torch, da, _ffi, ctx, hs, buf, info, rounds=5)
        for k in e["kernels"]:
            k["frac_of_8TBps"] = k["GBps"] / 8000.0
        encode_kernels = dict(e, note="secondary figure, not `value`: COPY TO's Vector -> Arrow encode (ArrowAppender semantics) of the "
                                      "same table, kernels only; offsets and data buffers of the first record batch compared with the "
                                      "source stream's (payload_matches_source); rocprofv3 stats + FETCH/WRITE passes of "
                                      "tools/encode_bench.py: profiles/r02/encode/")
    hs.close()
    del hs

    # ---- secondary, never `value`: the scan OPERATOR over files (BASELINE config 3) ----
    operator_path = None
    if not args.no_operator_path:
        # one directory per run, the same name on every rank: the self-launcher's run id, else the launcher's rendezvous port
        run_id = os.environ.get("MI_BENCH_RUN_ID") or ("port%s" % os.environ.get("MASTER_PORT", "0") if world > 1 else str(os.getpid()))
        d = os.path.join(args.shm_dir, "mi_bench_%s" % run_id)
        n_files = 8
        paths = [os.path.join(d, "lineitem_%d.arrows" % i) for i in range(n_files)]
        try:
            if rank == 0:   # rank 0's shard IS the SF(sf) table (first_row 0): written once, read by every rank
                os.makedirs(d, exist_ok=True)
                offs, nb = info["batch_offsets"], info["n_batches"]
                per = (nb + n_files - 1) // n_files
                for i, p in enumerate(paths):
                    lo, hi = offs[min(nb, i * per)], offs[min(nb, (i + 1) * per)]
                    with open(p, "wb") as f:
                        f.write(buf[: offs[0]].tobytes())
                        f.write(memoryview(buf[lo:hi]))
                        f.write(b"\xff\xff\xff\xff\x00\x00\x00\x00")
            barrier()
            con = da.Connection(local_rank)
            legs = {}
            for leg, kw, flt in (("full_scan_host_consumer", {}, False),
                                 ("config3_shipdate_pushdown", {"filter_compact": True}, True)):
                best = None
                for _ in range(2):   # first pass warms the page cache mappings and the pinned rings
                    rel = con.read_arrow(paths, rank=rank, world=world, **kw)
                    if flt:
                        rel.filter_range("l_shipdate", SHIP_LO, SHIP_HI)
                    barrier()
                    t1 = time.perf_counter()
                    got = rel.count(detail=True)
                    dt = time.perf_counter() - t1
                    scan_stats = rel.stats()
                    rel.close()
                    dt = max_over_ranks(dt)
                    cnt = torch.tensor([got["rows"], got["selected"]], dtype=torch.int64, device="cuda" if (world > 1 and args.backend == "nccl") else "cpu")
                    if world > 1:
                        dist.all_reduce(cnt)
                    best = dt if best is None else min(best, dt)
                    rows_all, sel_all = int(cnt[0].item()), int(cnt[1].item())
                assert rows_all == info["n_rows"], (rows_all, info["n_rows"])
                mine = max(1, got["rows"])   # PCIe bytes per row of THIS rank's share (mi_scan_get_stats)
                legs[leg] = {"seconds": best, "rows_per_s": rows_all / best, "rows": rows_all, "selected": sel_all,
                             "file_GBps": sum(os.path.getsize(p) for p in paths) / best / 1e9,
                             "h2d_bytes_per_row": scan_stats["h2d_bytes"] / mine, "d2h_bytes_per_row": scan_stats["d2h_bytes"] / mine,
                             "aliased_bytes_per_row": scan_stats["aliased_bytes"] / mine}
            if world == 1 and not args.no_cpu_baseline:
                # the CPU beside the multi-file legs (min(files, cores) threads, one file per thread -- the reference
                # scans one file per thread, src/file_scanner/arrow_file_scan.cpp:35-42): the oracle port over the same 8 files
                try:
                    legs["cpu_baseline_multifile"] = cpu_multifile_baseline(paths)
                except Exception as e:   # a secondary figure must not cost the line
                    legs["cpu_baseline_multifile"] = {"error": repr(e)[:200]}
            if world == 1:
                # BASELINE config 4 through the operator: COPY (FROM read_arrow(files)) TO 'out.arrows' (row_group_size 122880);
                # decode and encode both on the GPU, only the finished IPC bodies travel back (mi_writer_sink_scan)
                opath = os.path.join(d, "copy_out.arrows")
                for target, leg in ((opath, "config4_copy_to_file"), ("/dev/null", "config4_copy_to_null_sink")):
                    best = None
                    for _ in range(2):
                        if target == opath and os.path.exists(opath):
                            os.remove(opath)
                        t1 = time.perf_counter()
                        con.copy_to(con.read_arrow(paths), target, row_group_size=122880)
                        dt = time.perf_counter() - t1
                        best = dt if best is None else min(best, dt)
                    legs[leg] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "GBps_out": float(buf.size) / best / 1e9}
                if os.path.exists(opath):
                    os.remove(opath)
                # the same table as ONE compressed stream -- LZ4_FRAME (what pyarrow / Feather V2 write by default) and
                # ZSTD (the codec the reference registers a decompressor for and its benchmark writes, benchmark/lineitem.py:135)
                # -- device-resident count: bodies decompressed by the reader's host threads vs in HBM by the K8 kernels (the
                # compressed bytes cross PCIe), in the default environment.
                try:
                    import subprocess
                    import pyarrow as pa
                    import pyarrow.ipc as ipc
                    best = None
                    for _ in range(2):
                        rel = con.read_arrow(paths, device_resident=True, pipeline_depth=8)
                        t1 = time.perf_counter()
                        rel.count(detail=True)
                        dt = time.perf_counter() - t1
                        rel.close()
                        best = dt if best is None else min(best, dt)
                    plain = {"seconds": best, "rows_per_s": info["n_rows"] / best}
                    for codec, depth in (("lz4", 8), ("zstd", 16)):
                        cpath = os.path.join(d, "lineitem_%s.arrows" % codec)
                        reader = ipc.open_stream(pa.py_buffer(buf))
                        with ipc.new_stream(cpath, reader.schema, options=ipc.IpcWriteOptions(compression=codec)) as w:
                            for b in reader:
                                w.write_batch(b)
                        leg = {"file_bytes": os.path.getsize(cpath), "pipeline_depth": depth, "uncompressed_files": plain}
                        for tag, kw in (("host_threads", {"host_decompress": True}), ("in_hbm", {"host_decompress": "gpu"}),
                                        ("auto", {})):
                            best = None
                            for _ in range(2):
                                rel = con.read_arrow(cpath, device_resident=True, pipeline_depth=depth, **kw)
                                t1 = time.perf_counter()
                                got = rel.count(detail=True)
                                dt = time.perf_counter() - t1
                                st = rel.stats()
                                rel.close()
                                assert got["rows"] == info["n_rows"]
                                best = dt if best is None else min(best, dt)
                            leg[tag] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "h2d_bytes": st["h2d_bytes"],
                                        "batches_decompressed_in_hbm": st["lz4_batches_on_device"] + st["zstd_batches_on_device"]}
                        os.remove(cpath)
                        # the library asks for 24 hardware queues when it is loaded before the process's first HIP call and nobody
                        # has set GPU_MAX_HW_QUEUES (c_api.cpp): this process runs that way.  The HIP runtime's own default (4),
                        # for contrast, in a process of its own
                        leg["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES", "unset: the library's 24")
                        if codec == "zstd":
                            env = dict(os.environ, GPU_MAX_HW_QUEUES="4")
                            run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lz4_bench.py"), "--codec", codec, "--sf", str(args.sf),
                                                  "--dir", args.shm_dir, "--depth", str(depth), "--legs", "lz4_in_hbm"], env=env, capture_output=True,
                                                 text=True, timeout=600)
                            if run.returncode == 0:
                                z = json.loads(run.stdout.strip().split("\n")[-1])
                                leg["with_4_hw_queues"] = {"GPU_MAX_HW_QUEUES": 4, "in_hbm": {k: z["lz4_in_hbm"][k] for k in ("seconds", "rows_per_s")}}
                            else:
                                leg["with_4_hw_queues"] = {"error": run.stderr[-300:]}
                        legs["%s_device_resident_scan" % codec] = leg
                except ImportError:
                    pass   # no pyarrow on this box: the leg needs it to write the compressed stream
            con.close()
            operator_path = dict(legs, scaling="strong", files=n_files, rows=legs["full_scan_host_consumer"]["rows"], table="TPC-H SF%g lineitem (%d rows) in %s" % (args.sf, info["n_rows"], args.shm_dir),
                                 rows_per_s=legs["full_scan_host_consumer"]["rows_per_s"],
                                 note="secondary figures, never `value`: PCIe / page-cache inclusive; one table shared by all ranks, every "
                                      "rank takes the record batches k with k mod world == rank (mi_scan_options.rank / world)")
            barrier()
        finally:
            if rank == 0:
                for p in paths:
                    if os.path.exists(p):
                        os.remove(p)
                if os.path.isdir(d):
                    os.rmdir(d)

    sf100 = None
    if rank == 0 and world == 1 and args.sf == 10.0 and not args.rows and not args.no_sf100:
        sf100 = run_sf100_child(args, torch)
    if rank == 0:
        total_rows = info["n_rows"] * world
        ms_per_step = elapsed / args.steps * 1e3
        rows_per_s = total_rows * args.steps / elapsed
        alg_bytes = stats["bytes_read"] + stats["bytes_written"]
        kernels = kernel_table(cstats, per_class, pmc, same_workload)
        out = {
            "metric": "rows/sec + achieved HBM GB/s, TPC-H lineitem.arrows scan at 1/2/4/8 GPUs",
            "value": rows_per_s,
            "unit": "rows/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32/int64 (byte and integer transcode, no FP)",
            "data": "synthetic",
            "host_threads_on_gpu_numa_node": numa_node,   # -1: not bound (--no-numa-bind, or the platform names no node)
            "config": {"workload": "TPC-H SF%g lineitem.arrows full-column scan per GPU (%d rows, %d record batches of 122880, "
                                   "16 columns, validity bitmaps %s), IPC bodies resident in HBM, every vector materialised"
                                   % (args.sf, info["n_rows"], info["n_batches"], "absent" if args.no_validity else "present"),
                       "rows_per_gpu": info["n_rows"], "record_batches_per_gpu": info["n_batches"],
                       "sharding": "row groups, no collective"},
            "achieved_hbm_GBps_whole_step": alg_bytes * world / (elapsed / args.steps) / 1e9,
            "algorithmic_bytes_per_row": {"read": stats["bytes_read"] / info["n_rows"], "written": stats["bytes_written"] / info["n_rows"]},
            "roofline": roofline_of(kernels, alg_bytes, ms_per_step, traffic_stale),
            "north_star_read_roofline": read_roofline(rows_per_s / world, stats["bytes_read"] / info["n_rows"], stats["bytes_written"] / info["n_rows"]),
            "kernels": kernels,
            "cpu_baseline": cpu_baseline,
            "cpu_baseline_encode": cpu_baseline_encode,
            "parity": parity,
            "reference_shaped": ref_shaped,
            "config4_encode_kernels": encode_kernels,
            "operator_path": operator_path,
            "sf100": sf100,
            "setup_seconds": {"generate": t_gen, "parse_upload_plan": t_upload},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
