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
         os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_stream_reader.cpp"), "-ldl", "-lpthread", "-o", exe],
        capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    files = [os.path.join(golden_dir, f) for f in ("edge_nested.arrows", "ref_data/test.arrows", "edge_dict.arrows",
                                                   "edge_file_format.arrow", "edge_types2.arrows", "edge_empty.arrows")]
    run = subprocess.run([exe, os.environ.get("MI_SANITIZE_ITERS", "400")] + files, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", MI_IO_THREADS="2"))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "no sanitizer report" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
