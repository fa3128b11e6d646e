#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine_greedy.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine_greedy.log | tail -5
tail -25 gpurun_out/pytest_engine_greedy.log | cut -c1-300
bash scripts/gpu_ab_env.sh ${1:-2}
