#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ab_env.log
source <(sed -n '/^run() {/,/^}/p' scripts/gpu_ab_env.sh)
for rep in 1 2 3; do
  run "balance (phase duration)" MI_GRAPH_STEPS=1 --
  run "no balance" MI_GRAPH_STEPS=1 MI_ENGINE_BALANCE=0 --
done
