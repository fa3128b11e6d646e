#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for knobs in "1 3" "0 3" "1 2" "0 2"; do
  set -- $knobs
  echo "== thin=$1 depth=$2"
  MI_ENGINE_THIN=$1 MI_ENGINE_DEPTH=$2 timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done 2>&1 | tee gpurun_out/knobs.log
