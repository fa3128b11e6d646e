#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine.log | tail -3
bash scripts/gpu_ab.sh ${1:-2}
