#!/bin/bash
# A/B of engine build variants on ONE box (box-to-box spread is ~1-2 %, larger than most single changes):
#   here:  python scripts/build_variants.py
#   then:  gpurun --timeout 900 -- 'bash scripts/gpu_ab.sh [reps] [extra bench args]'
# Every variant in mistral-inference_amd/lib/variants/ and the main library are benchmarked `reps` times, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
REPS=${1:-2}; shift
LOG=gpurun_out/ab.log
: > $LOG
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras $EXTRA > gpurun_out/v.out 2>&1
  python - "$label" <<'PY' | tee -a gpurun_out/ab.log
import json, sys
try:
    d = json.loads(open('gpurun_out/v.out').read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f}  tok/s {d['value']:.1f}  step frac {d['hbm_roofline_step']['frac']:.4f}  kernel us {r.get('avg_launch_us')}  prefill tok/s {d['prefill']['tokens_per_s']}  [{d['config']['decode_launch']}]")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/v.out').read()[-400:])
PY
}
EXTRA="$*"
for rep in $(seq 1 $REPS); do
  run main X=1
  for f in mistral-inference_amd/lib/variants/libmistral_hip_*.so; do
    [ -f "$f" ] || continue
    n=$(basename $f .so); n=${n#libmistral_hip_}
    run $n MISTRAL_HIP_LIB=$PWD/$f
  done
done
[ -n "$AB_EXTRAS" ] && EXTRA="$* --loop forward" run main_forward_loop X=1
[ -n "$AB_EXTRAS" ] && EXTRA="$* --no-graph" run main_no_graph X=1
