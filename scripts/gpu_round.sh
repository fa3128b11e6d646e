#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench.log 2>&1
MISTRAL_HIP_LIB=$REPO/mistral-inference_amd/lib/libmistral_hip_asm.so timeout 900 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench_asm.log 2>&1
tail -n 5 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench.log gpurun_out/bench_asm.log; do
grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'], 'TF', d.get('roofline',{}).get('achieved'))
"; done
