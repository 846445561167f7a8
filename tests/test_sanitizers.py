"""AddressSanitizer + UBSan over the host IPC reader (CPU build only; GPU sanitizers are not available on the pool):
tests/sanitize/fuzz_reader.cpp is built with g++ from the two host sources alone (no HIP) and drains mutated fixtures,
touching every byte of every buffer span the reader hands out."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_reader_is_clean_under_asan_and_ubsan(tmp_path, golden_dir):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_reader")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "sanitize", "fuzz_reader.cpp"),
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_format.cpp"),
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_stream_reader.cpp"),
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "c_stream.cpp"), "-ldl", "-lpthread", "-o", exe],
        capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    files = [os.path.join(golden_dir, f) for f in ("edge_nested.arrows", "ref_data/test.arrows", "edge_dict.arrows",
                                                   "edge_file_format.arrow", "edge_types2.arrows", "edge_empty.arrows")]
    # + ZSTD / LZ4 bodies, stream and file format (written here by pyarrow)
    import numpy as np
    import pyarrow as pa
    import pyarrow.ipc as ipc
    rng = np.random.default_rng(5)
    t = pa.table({"a": rng.integers(0, 50, 12000), "s": ["row %d" % (i % 97) for i in range(12000)],
                  "l": pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(12000)], pa.list_(pa.int32()))})
    for codec in ("zstd", "lz4"):
        p1, p2 = str(tmp_path / ("c_%s.arrows" % codec)), str(tmp_path / ("c_%s.arrow" % codec))
        with ipc.new_stream(p1, t.schema, options=ipc.IpcWriteOptions(compression=codec)) as w:
            w.write_table(t, max_chunksize=5000)
        with ipc.new_file(p2, t.schema, options=ipc.IpcWriteOptions(compression=codec)) as w:
            w.write_table(t, max_chunksize=5000)
        files += [p1, p2]
    # a flat LZ4 table whose buffers span several linked 64 KiB blocks: these record batches are DEFERRED (handed out
    # compressed with the block tables of the GPU decompressor); the harness restates the K8 kernels and compares with liblz4
    flat = pa.table({"k": np.arange(60000, dtype=np.int64) * 7, "z": np.zeros(60000, np.int32), "r": rng.integers(0, 1 << 60, 60000),
                     "s": ["comment %d %s" % (i % 311, "lorem ipsum"[: i % 11]) for i in range(60000)]})
    p3 = str(tmp_path / "flat_lz4.arrows")
    with ipc.new_stream(p3, flat.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(flat, max_chunksize=40000)
    files.append(p3)
    run = subprocess.run([exe, os.environ.get("MI_SANITIZE_ITERS", "400")] + files, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", MI_IO_THREADS="2",
                                  TMPDIR=str(tmp_path)))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    import re
    assert int(re.search(r"(\d+) deferred LZ4 batches", run.stdout).group(1)) >= 2, run.stdout
    assert "no sanitizer report" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr


def test_zstd_stages_on_the_cpu(tmp_path):
    """The entropy stage of the ZSTD kernels (duckdb-arrow_amd/csrc/zstd_format.hpp: FSE / Huffman tables, backward
    bitstreams, sequences, repeat offsets) and the host walk that feeds it are plain C++ shared with the device build:
    tests/sanitize/zstd_check.cpp runs them block by block in the kernels' order on frames written here by libzstd (through
    pyarrow) and compares with the bytes that went in -- under ASan + UBSan, so an out-of-range table index shows here and not
    as a GPU fault."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import numpy as np
    import pyarrow as pa
    exe = str(tmp_path / "zstd_check")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "sanitize", "zstd_check.cpp"),
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_format.cpp"),
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_stream_reader.cpp"), "-ldl", "-lpthread", "-o", exe],
        capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    rng = np.random.default_rng(7)
    words = [b"carefully", b"final", b"deposits", b"furiously", b"quickly", b"express", b"packages", b"sleep", b"blithely", b"regular"]
    text = b" ".join(words[i] for i in rng.integers(0, len(words), 60000))
    cases = {
        "empty": b"", "one": b"x", "zeros": bytes(300000),
        "rand_small": rng.integers(0, 256, 1000, dtype=np.uint8).tobytes(),
        "rand_big": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),                       # raw blocks
        "prices": rng.integers(90000, 10500000, 100000).astype(np.int64).tobytes(),
        "dates": np.sort(rng.integers(8000, 10600, 200000).astype(np.int32)).tobytes(),
        "text": text,
        "offsets": np.cumsum(rng.integers(10, 44, 200000)).astype(np.int32).tobytes(),
        "flags": rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), 300000).tobytes(),            # 2-bit alphabet: direct weights
        "period": b"abcdefg" * 60000,
        "few": rng.choice(np.frombuffer(b"ab", dtype=np.uint8), 3000, p=[0.9, 0.1]).tobytes(),
        "mixed": text[:150000] + rng.integers(0, 256, 50000, dtype=np.uint8).tobytes() + bytes(70000) + text[:90000],
    }
    args = []
    for name, data in cases.items():
        raw = str(tmp_path / (name + ".raw"))
        open(raw, "wb").write(data)
        for level in (1, 3, 19):
            z = str(tmp_path / ("%s_%d.zst" % (name, level)))
            open(z, "wb").write(pa.Codec("zstd", compression_level=level).compress(data, asbytes=True))
            args += [z, raw]
    run = subprocess.run([exe] + args, capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "%d frames, 0 failed" % (len(args) // 2) in run.stdout, run.stdout[-2000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
    # every table mode and literal type the format has was seen (the first printed line counts them)
    import re
    seen = [int(x) for x in re.findall(r"\d+", run.stdout.split("\n")[0])]
    assert all(v > 0 for v in seen[:3]), run.stdout
