#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine.log | tail -3
: > gpurun_out/ab_env.log
source <(sed -n '/^run() {/,/^}/p' scripts/gpu_ab_env.sh)
run "nemo 40 layers" X=1 -- --model nemo-12b --prefill 8192 --steps 16 --warmup 3
run "nemo 40 layers, launch path" MI_DECODE_ENGINE=0 -- --model nemo-12b --prefill 8192 --steps 16 --warmup 3
run "7B" X=1 --
run "7B launch path" MI_DECODE_ENGINE=0 --
