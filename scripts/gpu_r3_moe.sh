#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine_greedy.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine_greedy.log | tail -5
tail -25 gpurun_out/pytest_engine_greedy.log | cut -c1-300
for rep in 1 2; do
timeout 600 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_mixtral8x7b.json.log; cut -c1-1500 gpurun_out/bench_mixtral8x7b.json.log
MI_DECODE_ENGINE=0 timeout 600 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
done
timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
