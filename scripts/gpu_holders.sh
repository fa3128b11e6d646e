#!/bin/bash
# A/B of the holder waves (round 2 ended with ONE run of scripts/holders_quick.py: bit-equal, 2.664 ms per eager step):
#   here, before the gpurun call:
#     (cd mistral-inference_amd && python -c "import build_native as b; b.build(); \
#        b.build(extra_flags=('-DENG_HOLDERS=0',), obj_dir='/tmp/obj_h0', lib='lib/variants/libmistral_hip_holders0.so')")
#   then:  gpurun --timeout 600 -- 'bash scripts/gpu_holders.sh'
# Expect: engine tests bit-equal (9 passed); main (3 holder waves) faster than the holders0 variant by ~3 us per layer.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/holders.log
: > $LOG
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $LOG
run() { echo "== $*" | tee -a $LOG; env "$@" timeout 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/v.out 2>&1; python -c "import json; d=json.loads(open('gpurun_out/v.out').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1 | tee -a $LOG; }
for rep in 1 2 3; do
  run X=main
  run MISTRAL_HIP_LIB=$PWD/mistral-inference_amd/lib/variants/libmistral_hip_holders0.so
done
timeout 300 python scripts/engine_trace.py 2>&1 | tail -45 | tee -a $LOG
