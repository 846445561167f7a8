#!/bin/bash
# The profile recipe of one round, run ON the GPU box (through gpurun):  bash tools/profile_round.sh <tag>
# Three separate rocprofv3 runs of the SAME bench command: kernel stats, then one PMC pass per counter (counters are
# never combined with hip/hsa/sys tracing).  Outputs land under gpurun_out/<tag>_*; tools/profile_summary.py turns
# them into profiles/<tag>/.
set -eo pipefail
tag=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o runc --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -o runc --output-format csv -- python3 $BENCH > gpurun_out/${tag}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_write -o runc --output-format csv -- python3 $BENCH > gpurun_out/${tag}_write.log 2>&1
cat gpurun_out/${tag}_bench.json
