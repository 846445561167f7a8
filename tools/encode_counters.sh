#!/bin/bash
# Where the waves of encode_string_1p spend their cycles: SQ counters in separate rocprofv3 passes (never combined with
# tracing domains other than the kernel trace).  Run ON the GPU box: bash tools/encode_counters.sh <tag>
set -eo pipefail
tag=${1:-r02_enc_sq}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/${tag}_$c -o run --output-format csv -- python3 tools/encode_bench.py --sf 10 --rounds 2 > gpurun_out/${tag}_$c.log 2>&1 || echo "$c failed"
done
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(dict)
for path in glob.glob("gpurun_out/${tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        name = "encode_string_1p" if "encode_string_1p" in k else "encode_fixed" if "encode_fixed" in k else None
        if name:
            out[name].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for name, cs in out.items():
    print(name, {c: sum(v) / len(v) for c, v in sorted(cs.items())})
PY
