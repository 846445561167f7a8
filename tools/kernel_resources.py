#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the gfx950 build (hipcc -Rpass-analysis=kernel-resource-usage)."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "duckdb-arrow_amd", "csrc")
for f in sorted(glob.glob(os.path.join(src, "kernels_*.hip"))):
    if len(sys.argv) > 1 and sys.argv[1] not in f:
        continue
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", f, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=src).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = re.sub(r"\(.*", "", cur["name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("miarrow::device::", "")
                print("%-34s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-5s occupancy %-2s LDS %s" % (
                    name[:34], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize [bytes/lane]"),
                    cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
