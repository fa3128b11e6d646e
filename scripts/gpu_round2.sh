#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/round2.log
echo "== engine trace" | tee $LOG
timeout 600 python scripts/engine_trace.py 2>&1 | tail -45 | tee gpurun_out/engine_trace.log | tee -a $LOG
echo "== bench engine" | tee -a $LOG
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $LOG
echo "== bench launches" | tee -a $LOG
MI_DECODE_ENGINE=0 timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee -a $LOG
if [ "${1:-}" = "tests" ]; then
echo "== tests" | tee -a $LOG
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 | tee -a $LOG
fi
