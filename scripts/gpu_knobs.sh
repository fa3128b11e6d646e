#!/bin/bash
# A/B of the engine's runtime knobs on one box.  MI_ENGINE_THIN = 0 stream / 1 thin / 2 stop during sweeps; MI_ENGINE_DEPTH 2|3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/knobs.log
: > $LOG
run() { echo "== $*" | tee -a $LOG; env "$@" timeout 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/v.out 2>&1; python -c "import json; d=json.loads(open('gpurun_out/v.out').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1 | tee -a $LOG; }
for rep in 1 2; do
  for t in ${KNOBS:-0 1 2}; do run MI_ENGINE_THIN=$t; done
  run MI_ENGINE_THIN=2 MI_ENGINE_DEPTH=3
done
