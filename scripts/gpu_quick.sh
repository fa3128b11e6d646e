#!/bin/bash
# quick engine check: bit-equality tests, timeline, A/B bench (no full suite)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/quick.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | tail -3 | tee $LOG
timeout 600 python scripts/engine_trace.py 2>&1 | tail -34 | tee -a $LOG
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $LOG
