"""sha256 over the kernel sources (csrc/kernels_*.hip, *.inl, device_common.hpp): tools/profile_summary.py stamps the PMC
traffic summary with it, bench.py compares and reports `traffic_stale` when the kernels have changed since the counters
were taken."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    d = os.path.join(ROOT, "duckdb-arrow_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "kernels_*.hip")) + glob.glob(os.path.join(d, "*.inl")) + [os.path.join(d, "device_common.hpp")])
    h = hashlib.sha256()
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha())
