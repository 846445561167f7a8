#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 200 python tools/_queues_ab.py > gpurun_out/r03_qab.txt 2>/dev/null || exit 1
timeout -k 10 200 python tools/_queues_ab.py --torch >> gpurun_out/r03_qab.txt 2>/dev/null || exit 2
GPU_MAX_HW_QUEUES=24 timeout -k 10 200 python tools/_queues_ab.py --torch >> gpurun_out/r03_qab.txt 2>/dev/null || exit 3
GPU_MAX_HW_QUEUES=32 timeout -k 10 200 python tools/_queues_ab.py --torch >> gpurun_out/r03_qab.txt 2>/dev/null || exit 4
GPU_MAX_HW_QUEUES=32 timeout -k 10 200 python tools/_queues_ab.py >> gpurun_out/r03_qab.txt 2>/dev/null || exit 5
cat gpurun_out/r03_qab.txt
