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
  env "${envs[@]}" timeout 600 python bench.py $defaults --no-cpu-baseline --no-extras "$@" > gpurun_out/v.out 2>&1
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
  run "shipped (1 token per launch, graph)" X=1 --
  run "8 steps per hipGraph" MI_GRAPH_STEPS=8 --
  run "driver shape: steps 20 warmup 5" X=1 -- --steps 20 --warmup 5
done
run "forward() + torch.argmax loop" X=1 -- --loop forward
run "launch path (engine off)" MI_DECODE_ENGINE=0 --
