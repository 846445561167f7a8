"""bench.py's launch contract: `python bench.py --gpus N` without a launcher starts the N ranks itself (the parent never
touches the GPU), reports n_gpus = the ranks that ran, and fails loudly when a rank fails.  The unit the ranks share is the
record batch (the reference's is the file: one reader thread per file, src/file_scanner/arrow_file_scan.cpp:35-42)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_a_failing_rank_fails_the_run():
    """No GPU in the CPU container: every rank dies at its first device call, and the parent -- which made none -- says so
    with a non-zero exit code instead of printing a number."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    run = _bench("--gpus", "2", "--backend", "gloo", "--rows", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-operator-path")
    assert run.returncode != 0
    assert "ranks failed" in run.stderr
    assert not [ln for ln in run.stdout.split("\n") if ln.startswith("{")]


@pytest.mark.gpu
def test_self_launcher_runs_two_ranks_on_one_device():
    """Two ranks rehearsed on ONE device over gloo: both shards bit-exact against the oracle, the operator-path table scanned
    exactly once between them (record batch k -> rank k mod 2), n_gpus = the ranks that ran."""
    run = _bench("--gpus", "2", "--single-device", "--backend", "gloo", "--rows", "300000", "--steps", "2", "--no-cpu-baseline")
    assert run.returncode == 0, run.stderr[-2000:]
    z = json.loads([ln for ln in run.stdout.split("\n") if ln.startswith("{")][-1])
    assert z["n_gpus"] == 2 and z["scaling"] == "weak"
    assert z["config"]["rows_per_gpu"] == 300000
    assert z["parity"]["bit_exact"] and z["parity"]["ranks_bit_exact"] == 2
    op = z["operator_path"]
    assert op["rows"] == 300000 and op["full_scan_host_consumer"]["rows"] == 300000
    assert op["config3_shipdate_pushdown"]["rows"] == 300000 and 0 < op["config3_shipdate_pushdown"]["selected"] < 300000
    assert z["roofline"]["frac"] > 0 and z["sf100"] is None
