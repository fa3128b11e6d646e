#!/bin/bash
# A/B of the engine's runtime knobs on one box.  MI_ENGINE_THIN bits: 1 thin / 2 stop the loader during sweeps, 4 arrival flags
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/knobs.log
: > $LOG
run() { echo "== $*" | tee -a $LOG; env "$@" timeout 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/v.out 2>&1; tail -1 gpurun_out/v.out | cut -c1-170 | tee -a $LOG; }
for rep in 1 2; do
  for t in ${KNOBS:-2 6}; do
    run MI_ENGINE_THIN=$t
    python - <<'PY' || exit 1
import json,sys
try:
    ms=json.loads(open("gpurun_out/v.out").read().strip().splitlines()[-1])["ms_per_step"]
except Exception as e:
    print("no bench line", e); sys.exit(1)
sys.exit(0 if ms < 10 else 1)
PY
  done
done
MI_ENGINE_THIN=6 timeout 300 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | tail -3 | tee -a $LOG
