#!/bin/bash
# One GPU call for the decode engine: parity (engine == launch path, bit for bit), then A/B timing.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== engine tests (small shapes)" | tee gpurun_out/engine_round.log
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "not full_size" 2>&1 | tail -25 | tee -a gpurun_out/engine_round.log
echo "== bench engine ON" | tee -a gpurun_out/engine_round.log
MI_DECODE_ENGINE=1 timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_engine_on.log | tee -a gpurun_out/engine_round.log
echo "== bench engine OFF" | tee -a gpurun_out/engine_round.log
MI_DECODE_ENGINE=0 timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_engine_off.log | tee -a gpurun_out/engine_round.log
echo "== engine full-size test" | tee -a gpurun_out/engine_round.log
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "full_size" 2>&1 | tail -15 | tee -a gpurun_out/engine_round.log
