"""Mutated DATA buffers (offsets, views, dictionary indices, list offsets, bitmaps) through the scan operator on the GPU:
the kernels validate every data-dependent address before they dereference it (FULL validation on the device), so a
damaged body ends in a clean MiError or in (different) values -- never in a device fault or a host crash."""
import os

import numpy as np
import pytest

import duckdb_arrow_amd as da
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rel", ["edge_nested.arrows", "ref_data/test.arrows", "edge_dict.arrows", "edge_types2.arrows"])
def test_mutated_bodies_end_in_errors_not_faults(golden_dir, rel):
    src = np.fromfile(os.path.join(golden_dir, rel), np.uint8)
    msgs = po.walk_stream(src)
    bodies = [(m["body_off"], m["body_len"]) for m in msgs if m["body_len"] > 0]
    rng = np.random.default_rng(abs(hash(rel)) % 2**32)
    con = da.Connection(0)
    errors = clean = 0
    for it in range(int(os.environ.get("MI_DATA_FUZZ_ITERS", "60"))):
        buf = src.copy()
        for _ in range(int(rng.integers(1, 4))):
            off, ln = bodies[int(rng.integers(0, len(bodies)))]
            p = off + int(rng.integers(0, ln))
            if it % 3 == 0:      # a whole int32 / int64 replaced by an extreme value (offsets, indices, view fields)
                p = p // 8 * 8
                buf[p: p + 8] = np.frombuffer(np.int64(rng.choice([-1, 2**31 - 1, 2**40, -2**31, 2**62])).tobytes(), np.uint8)
            else:
                buf[p] = int(rng.integers(0, 256))
        try:
            rel_ = con.scan_arrow_ipc([buf], accept_dictionaries=True)
            cols = rel_.fetch_columns()
            assert len(cols) == len(rel_.columns)
            clean += 1
        except da.MiError:
            errors += 1
    con.close()
    assert errors + clean > 0
