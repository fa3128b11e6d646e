#!/bin/bash
# Round-3 first GPU call: engine + greedy tests, smoke, then the A/B of the engine build variants and one phase trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine_greedy.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine_greedy.log | tail -5
tail -30 gpurun_out/pytest_engine_greedy.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | cut -c1-400
bash scripts/gpu_ab.sh ${1:-2}
timeout 300 python scripts/engine_trace.py > gpurun_out/engine_trace.log 2>&1; tail -45 gpurun_out/engine_trace.log
