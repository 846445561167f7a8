#!/bin/bash
# The profile recipe of one round, run ON the GPU box (through gpurun):  bash tools/profile_round.sh <tag> [workloads...]
# Per workload three separate rocprofv3 runs of the SAME command: kernel stats, then one PMC pass per counter (counters are
# never combined with hip/hsa/sys tracing).  Outputs land under gpurun_out/<tag>_<workload>_*; tools/profile_summary.py
# turns them into profiles/<tag>/<workload>/.  Workloads: decode (bench.py, the headline), encode (tools/encode_bench.py,
# BASELINE config 4), commits (tools/commits_bench.py, config 5), filter (tools/filter_bench.py, K6 + gather), lz4
# (tools/lz4_bench.py, K8 through the scan operator), zstd (the same with --codec zstd, 16 slots).
set -eo pipefail
tag=${1:-r03}
shift || true
workloads=${@:-decode encode commits filter lz4}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for w in $workloads; do
  case $w in
    decode)  full="bench.py --no-operator-path"; short="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operator-path --no-encode-leg" ;;
    encode)  full="tools/encode_bench.py --sf 10 --per-column --native-heap"; short="tools/encode_bench.py --sf 10 --rounds 3" ;;
    commits) full="tools/commits_bench.py"; short="tools/commits_bench.py" ;;
    filter)  full="tools/filter_bench.py"; short="tools/filter_bench.py" ;;
    lz4)     full="tools/lz4_bench.py --sf 10 --legs plain,lz4_host_threads,lz4_in_hbm"; short="tools/lz4_bench.py --sf 2 --legs lz4_in_hbm" ;;
    zstd)    full="tools/lz4_bench.py --codec zstd --depth 16 --sf 10 --legs plain,lz4_host_threads,lz4_in_hbm"; short="tools/lz4_bench.py --codec zstd --depth 16 --sf 2 --legs lz4_in_hbm" ;;
    *) echo "unknown workload $w"; exit 2 ;;
  esac
  p=gpurun_out/${tag}_${w}
  timeout -k 10 400 python3 $full > ${p}.json 2> ${p}.err
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d ${p}_stats -o runc --output-format csv -- python3 $short > ${p}_stats.log 2>&1
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${p}_fetch -o runc --output-format csv -- python3 $short > ${p}_fetch.log 2>&1
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${p}_write -o runc --output-format csv -- python3 $short > ${p}_write.log 2>&1
  echo "== $w"; tail -c 600 ${p}.json; echo
done
