"""sha256 over the sources of the decode kernels (csrc/kernels_decode.hip, device_common.hpp -- what `roofline.traffic` in
bench.py's line was counted on): tools/profile_summary.py stamps the PMC traffic summary with it, bench.py compares and
reports `traffic_stale` when those kernels have changed since the counters were taken."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    d = os.path.join(ROOT, "duckdb-arrow_amd", "csrc")
    files = [os.path.join(d, "device_common.hpp"), os.path.join(d, "kernels_decode.hip")]
    h = hashlib.sha256()
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha())
