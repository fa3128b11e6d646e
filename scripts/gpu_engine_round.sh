#!/bin/bash
# One GPU call for the decode engine: parity (engine == launch path, bit for bit), timeline, A/B timing.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== engine trace" | tee gpurun_out/engine_round.log
timeout 600 python scripts/engine_trace.py 2>&1 | tail -60 | tee gpurun_out/engine_trace.log | tee -a gpurun_out/engine_round.log
echo "== bench engine ON" | tee -a gpurun_out/engine_round.log
MI_DECODE_ENGINE=1 timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_engine_on.log | tee -a gpurun_out/engine_round.log
if [ "${1:-}" = "tests" ]; then
echo "== engine tests" | tee -a gpurun_out/engine_round.log
timeout 1200 python -m pytest tests/test_gpu_engine.py -q 2>&1 | tail -40 | tee -a gpurun_out/engine_round.log
fi
