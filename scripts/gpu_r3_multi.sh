#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x > gpurun_out/pytest_engine_greedy.log 2>&1
grep -E "passed|failed|rror" gpurun_out/pytest_engine_greedy.log | tail -5
tail -25 gpurun_out/pytest_engine_greedy.log | cut -c1-300
: > gpurun_out/ab_env.log
source <(sed -n '/^run() {/,/^}/p' scripts/gpu_ab_env.sh)
for rep in 1 2; do
  run "32 tokens per launch" X=1 --
  run "1 token per launch (graph)" MI_LAUNCH_STEPS=1 --
  run "32/launch, steps 20 warmup 5" X=1 -- --steps 20 --warmup 5
  run "1/launch, steps 20 warmup 5" MI_LAUNCH_STEPS=1 -- --steps 20 --warmup 5
done
run "mixtral 32/launch" X=1 -- --model mixtral-8x7b --steps 32 --warmup 4
run "mixtral 1/launch" MI_LAUNCH_STEPS=1 -- --model mixtral-8x7b --steps 32 --warmup 4
