#!/bin/bash
# A/B of run-time switches of the decode loop on ONE box: each line of the table below is benchmarked REPS times, interleaved.
#   gpurun --timeout 900 -- 'bash scripts/gpu_ab_env.sh 2'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
REPS=${1:-2}
LOG=gpurun_out/ab_env.log
: > $LOG
run() {  # label, then env assignments, then -- bench args
  local label=$1; shift
  local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local defaults="--steps 64 --warmup 8"; case " $* " in *" --steps "*) defaults="";; esac
  env "${envs[@]}" timeout 600 python bench.py $defaults --no-cpu-baseline "$@" > gpurun_out/v.out 2>&1
  python - "$label" <<'PY' | tee -a gpurun_out/ab_env.log
import json, sys
try:
    d = json.loads(open('gpurun_out/v.out').read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print(f"{sys.argv[1]:34s} ms/step {d['ms_per_step']:.4f}  tok/s {d['value']:.1f}  step frac {d['hbm_roofline_step']['frac']:.4f}  kernel us {r.get('avg_launch_us')}  [{d['config']['decode_launch']}]")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/v.out').read()[-600:])
PY
}
for rep in $(seq 1 $REPS); do
  run "shipped (graph x8, balance)" X=1 --
  run "graph x1, balance" MI_GRAPH_STEPS=1 --
  run "graph x8, no balance" MI_ENGINE_BALANCE=0 --
  run "graph x1, no balance" MI_GRAPH_STEPS=1 MI_ENGINE_BALANCE=0 --
done
run "graph x32, balance" MI_GRAPH_STEPS=32 --
run "forward() + argmax loop" MI_ENGINE_BALANCE=0 -- --loop forward
