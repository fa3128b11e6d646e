#!/bin/bash
# round 4, call 1: the new tests + a driver-style bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/call1.log
: > $L
timeout 600 python -m pytest tests/test_gpu_sampling.py -q -x 2>&1 | tail -15 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_greedy.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -8 | tee -a $L
timeout 900 python -m pytest tests/test_gpu_depth.py -q -x -s -k "nemo or 8x22b" 2>&1 | grep -E "passed|failed|rror|Nemo|8x22B|assert" | cut -c1-600 | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r04_call1_bench_steps20.json.log | cut -c1-1500 | tee -a $L
