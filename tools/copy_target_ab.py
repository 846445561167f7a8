#!/usr/bin/env python3
"""What bounds `COPY ... TO 'file.arrows'` on this box once decode and encode run on the GPU (bench.py's config4_copy_to_file):
how fast can ~10 GB of finished IPC bodies that sit in HBM reach ONE fresh file in /dev/shm?  Variants, each moving the same
number of bytes in record-batch sized pieces (the COPY pump's row groups):

  write_hot        one thread, write() of a cache-hot host buffer (the floor a single writer has)
  write_after_d2h  one thread, D2H into pinned memory, then write() of it (what the fused pump's I/O thread does)
  pwrite_T         T threads, pwrite() into disjoint ranges of the file (the inode lock of tmpfs serialises them)
  fallocate_write  fallocate() the whole file first (timed), then one thread write()s
  map_register     ftruncate + mmap(MAP_SHARED) + hipHostRegister of the mapping in chunks by T threads (the pages are
                   allocated and pinned here), then D2H straight into the page cache: no CPU copy at all
  map_memcpy       the same mapping filled by memcpy from pinned memory (page faults under the copy)

usage: python tools/copy_target_ab.py [--gb 10.4] [--piece-mb 21.5] [--dir /dev/shm]"""
import argparse
import ctypes as C
import json
import mmap
import os
import sys
import threading
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=10.4)
    ap.add_argument("--piece-mb", type=float, default=21.5)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipDeviceSynchronize.argtypes = []
    D2H = 2
    piece = int(args.piece_mb * (1 << 20)) // 4096 * 4096
    n_pieces = max(1, int(args.gb * 1e9) // piece)
    total = piece * n_pieces
    dev = torch.randint(0, 255, (piece,), dtype=torch.uint8, device="cuda")          # one finished body in HBM
    pinned = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2)]
    hot = np.frombuffer(os.urandom(1 << 20) * (piece >> 20) + b"\0" * (piece - ((piece >> 20) << 20)), np.uint8)
    path = os.path.join(args.dir, "mi_copy_ab_%d.bin" % os.getpid())
    out = {"bytes": total, "piece_bytes": piece, "pieces": n_pieces, "dir": args.dir}
    only = set(x for x in args.only.split(",") if x)

    def fresh():
        if os.path.exists(path):
            os.remove(path)
        return os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)

    def record(name, seconds, **kw):
        out[name] = dict(seconds=seconds, GBps=total / seconds / 1e9, **kw)
        print(name, "%.3f s  %.2f GB/s" % (seconds, total / seconds / 1e9), kw, file=sys.stderr, flush=True)

    try:
        if not only or "write_hot" in only:
            fd = fresh()
            mv = memoryview(hot)
            t0 = time.perf_counter()
            for _ in range(n_pieces):
                os.write(fd, mv)
            record("write_hot", time.perf_counter() - t0)
            os.close(fd)
        if not only or "write_after_d2h" in only:
            fd = fresh()
            t0 = time.perf_counter()
            for i in range(n_pieces):
                p = pinned[i & 1]
                hip.hipMemcpy(p.data_ptr(), dev.data_ptr(), piece, D2H)
                os.write(fd, memoryview(p.numpy()))
            record("write_after_d2h", time.perf_counter() - t0, note="D2H and write() alternate on one thread")
            os.close(fd)
            # the pump's shape: D2H of piece i + 1 overlaps the write() of piece i
            fd = fresh()
            stream = torch.cuda.Stream()
            t0 = time.perf_counter()
            hip.hipMemcpyAsync(pinned[0].data_ptr(), dev.data_ptr(), piece, D2H, stream.cuda_stream)
            for i in range(n_pieces):
                stream.synchronize()
                if i + 1 < n_pieces:
                    hip.hipMemcpyAsync(pinned[(i + 1) & 1].data_ptr(), dev.data_ptr(), piece, D2H, stream.cuda_stream)
                os.write(fd, memoryview(pinned[i & 1].numpy()))
            record("write_after_d2h_overlapped", time.perf_counter() - t0)
            os.close(fd)
        for T in (2, 4):
            name = "pwrite_%d" % T
            if only and name not in only:
                continue
            fd = fresh()
            os.ftruncate(fd, total)
            mv = memoryview(hot)

            def work(t):
                for i in range(t, n_pieces, T):
                    os.pwrite(fd, mv, i * piece)
            ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            [x.start() for x in ts]
            [x.join() for x in ts]
            record(name, time.perf_counter() - t0)
            os.close(fd)
        if not only or "fallocate_write" in only:
            fd = fresh()
            t0 = time.perf_counter()
            os.posix_fallocate(fd, 0, total)
            t_alloc = time.perf_counter() - t0
            mv = memoryview(hot)
            t1 = time.perf_counter()
            for i in range(n_pieces):
                os.pwrite(fd, mv, i * piece)
            t_write = time.perf_counter() - t1
            record("fallocate_write", t_alloc + t_write, fallocate_seconds=t_alloc, write_seconds=t_write)
            os.close(fd)
        for T in (1, 2, 4):
            name = "map_register_%d" % T
            if only and name not in only:
                continue
            fd = fresh()
            os.ftruncate(fd, total)
            mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            base = C.addressof(C.c_char.from_buffer(mm))
            chunk = piece * 8
            chunks = [(o, min(chunk, total - o)) for o in range(0, total, chunk)]
            errs = []

            def reg(t):
                for o, ln in chunks[t::T]:
                    rc = hip.hipHostRegister(base + o, ln, 0)
                    if rc != 0:
                        errs.append(rc)
            ts = [threading.Thread(target=reg, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            [x.start() for x in ts]
            [x.join() for x in ts]
            t_reg = time.perf_counter() - t0
            if errs:
                out[name] = {"error": "hipHostRegister failed with %s" % sorted(set(errs))}
            else:
                t1 = time.perf_counter()
                for i in range(n_pieces):
                    hip.hipMemcpyAsync(base + i * piece, dev.data_ptr(), piece, D2H, None)
                hip.hipDeviceSynchronize()
                t_d2h = time.perf_counter() - t1
                record(name, t_reg + t_d2h, register_seconds=t_reg, d2h_seconds=t_d2h, register_GBps=total / t_reg / 1e9)
                t2 = time.perf_counter()
                for o, ln in chunks:
                    hip.hipHostUnregister(base + o)
                out[name]["unregister_seconds"] = time.perf_counter() - t2
            del base
            try:
                mm.close()
            except BufferError:
                pass
            os.close(fd)
        if not only or "map_memcpy" in only:
            fd = fresh()
            os.ftruncate(fd, total)
            mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            dst = np.frombuffer(mm, np.uint8)
            src = pinned[0].numpy()
            t0 = time.perf_counter()
            for i in range(n_pieces):
                dst[i * piece: (i + 1) * piece] = src
            record("map_memcpy", time.perf_counter() - t0)
            del dst
            mm.close()
            os.close(fd)
    finally:
        if os.path.exists(path):
            os.remove(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
