"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls: CPU only)."""
import os
import re
import subprocess

import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("mi_arrow_ipc.h", "mi_synth.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text))
    return names


def test_header_symbols_are_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (mi_[a-z0-9_]+)", out))
    missing = declared_symbols() - exported
    assert not missing, "declared in include/*.h but not exported: %s" % sorted(missing)


def test_binding_covers_every_declared_symbol():
    assert declared_symbols() == set(_ffi.SIGNATURES), set(_ffi.SIGNATURES) ^ declared_symbols()
    L = _ffi.lib()
    for name in _ffi.SIGNATURES:
        assert getattr(L, name) is not None


def test_duckdb_glue_calls_only_declared_and_exported_symbols():
    """glue/duckdb/*.{cpp,hpp} -- the extension's TableFunction / CopyFunction surface written against DuckDB's API, built only
    with -DDUCKDB_DIR -- may call nothing but what include/mi_arrow_ipc.h declares and the library exports; the sources name
    the reference file each of them replaces, and the build refuses to configure without a DuckDB tree (no stand-in headers)."""
    glue = os.path.join(ROOT, "glue", "duckdb")
    sources = sorted(f for f in os.listdir(glue) if f.endswith((".cpp", ".hpp")))
    assert {"mi_scan_arrow_ipc.cpp", "mi_read_arrow.cpp", "mi_write_arrow_stream.cpp", "mi_to_arrow_ipc.cpp",
            "mi_nanoarrow_extension.cpp", "mi_glue_common.hpp", "mi_file_scan.hpp"} <= set(sources)
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (mi_[a-z0-9_]+)", out))
    declared = declared_symbols()
    types = set(re.findall(r"\b(mi_[a-z0-9_]+)\b", " ".join(re.findall(r"typedef struct (\w+)|\} (\w+);", open(os.path.join(ROOT, "include", "mi_arrow_ipc.h")).read()).__str__().split())))
    called = set()
    for f in sources:
        text = open(os.path.join(glue, f)).read()
        assert "src/" in text, "%s does not cite the reference file it replaces" % f
        text = re.sub(r"//[^\n]*", "", text)
        called |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text))
    called -= types
    assert called and called <= declared, sorted(called - declared)
    assert called <= exported, sorted(called - exported)
    # the five entry points of the reference, under its names
    ext = open(os.path.join(glue, "mi_nanoarrow_extension.cpp")).read() + open(os.path.join(glue, "mi_read_arrow.cpp")).read() + \
        open(os.path.join(glue, "mi_scan_arrow_ipc.cpp")).read() + open(os.path.join(glue, "mi_write_arrow_stream.cpp")).read() + \
        open(os.path.join(glue, "mi_to_arrow_ipc.cpp")).read()
    for name in ('"nanoarrow_version"', '"read_arrow"', '"scan_arrow_ipc"', '"to_arrow_ipc"', 'CopyFunction function("arrows")', 'function.name = "arrow"'):
        assert name in ext, name
    cm = open(os.path.join(glue, "CMakeLists.txt")).read()
    assert "DUCKDB_DIR" in cm and "FATAL_ERROR" in cm
    assert not os.path.exists(os.path.join(glue, "duckdb")), "no stand-in DuckDB headers"


def test_struct_sizes_match_header():
    """ctypes mirrors vs the C compiler's view of include/mi_arrow_ipc.h."""
    src = r'''
#include <stdio.h>
#include "mi_arrow_ipc.h"
#include "mi_synth.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mi_batch_node), sizeof(mi_field), sizeof(mi_ipc_buffer), sizeof(mi_batch),
         sizeof(mi_batch_index_entry), sizeof(mi_col_task), sizeof(mi_scan_options), sizeof(mi_vector),
         sizeof(mi_data_chunk), sizeof(mi_write_options), sizeof(mi_synth_options), sizeof(mi_string_t));
  return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    import ctypes as C
    mine = [C.sizeof(x) for x in (_ffi.BatchNode, _ffi.Field, _ffi.IpcBuffer, _ffi.Batch, _ffi.BatchIndexEntry, _ffi.ColTask,
                                  _ffi.ScanOptions, _ffi.Vector, _ffi.DataChunk, _ffi.WriteOptions, _ffi.SynthOptions)] + [16]
    assert sizes == mine


def test_versions():
    assert da.nanoarrow_version() == "0.7.0-SNAPSHOT"  # test/sql/nanoarrow.test:15-18
    assert da.version().startswith("mi_arrow_ipc 2 gfx950")


def test_no_cpu_fallback_without_device():
    """The product path fails loudly when there is no HIP device."""
    if da.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(da.MiError) as e:
        da.Context(0)
    assert e.value.code == _ffi.MI_ENODEV and "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "duckdb-arrow_amd")
    for root, _, files in os.walk(pkg):
        if "build" in root.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="replace").read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle.h" not in text, os.path.join(root, f)


def _build_c_example(tmp_path, name="q6"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "duckdb-arrow_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", name + ".c"), "-L" + libdir, "-lmi_arrow_ipc", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True, capture_output=True)
    return exe


def test_plain_c_client_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/q6.c is C99 against include/mi_arrow_ipc.h alone (no C++, HIP or torch on the client side).  Without a
    device the first call reports MI_ENODEV -- the product path has no CPU fallback."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU variant")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([exe, os.path.join(root, "tests", "golden", "lineitem_sf0_01_q6.arrows")], capture_output=True, text=True)
    assert r.returncode == 1 and "mi_ctx_create failed (19)" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_client_computes_q6(tmp_path):
    """The same binary on a GPU box: TPC-H Q6 on lineitem SF0.01 = 1193053.2253 (test/nodejs/arrow_test.js:423-424)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([exe, os.path.join(root, "tests", "golden", "lineitem_sf0_01_q6.arrows")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "revenue = 1193053.2253  (1191 of 60175 rows pass)" in r.stdout


def test_hbm_example_builds(tmp_path):
    """examples/hbm_scan.c: the headline measurement from plain C99 (mi_hbm_* + mi_synth_*), -Werror clean."""
    _build_c_example(tmp_path, "hbm_scan")


@pytest.mark.gpu
def test_plain_c_client_runs_the_hbm_resident_scan(tmp_path):
    """The HBM-resident super-batch mode is reachable without Python: the C program plans, launches and times the scan of a
    small synthetic lineitem and reads a vector element back."""
    import subprocess
    exe = _build_c_example(tmp_path, "hbm_scan")
    r = subprocess.run([exe, "0.05", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "300060 rows in 3 messages, 48 tasks" in r.stdout and "transcode_string" in r.stdout and "l_orderkey[0] = " in r.stdout
