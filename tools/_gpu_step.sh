#!/bin/bash
# scratch: the GPU steps of the current measurement call (not part of the product; rewritten per call)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_zstd.py -m gpu -x -q > gpurun_out/r03_t13.log 2>&1; rc=$?; tail -12 gpurun_out/r03_t13.log
test $rc -eq 0 || exit 1
MI_ZSTD_PROBE=1 timeout -k 10 300 python tools/lz4_bench.py --codec zstd --depth 1 --sf 1 --legs lz4_in_hbm > gpurun_out/r03_zprobe3_d1.json 2> gpurun_out/r03_zprobe3_d1.err || exit 2
MI_SCAN_TRACE=1 timeout -k 10 300 python tools/lz4_bench.py --codec zstd --depth 16 --sf 10 --legs lz4_host_threads,lz4_in_hbm > gpurun_out/r03_zstd5_default.json 2> gpurun_out/r03_zstd5_default.err || exit 3
MI_SCAN_TRACE=1 timeout -k 10 300 python tools/lz4_bench.py --codec lz4 --depth 8 --sf 10 --legs plain,lz4_in_hbm > gpurun_out/r03_lz4_5.json 2> gpurun_out/r03_lz4_5.err || exit 5
MI_SCAN_TRACE=1 timeout -k 10 300 python tools/host_scan_bench.py > gpurun_out/r03_hostscan5.json 2> gpurun_out/r03_hostscan5.err || exit 6
echo all-ok
