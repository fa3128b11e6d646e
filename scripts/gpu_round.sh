#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/bench.log 2>&1
grep -h '"metric"' gpurun_out/bench.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'])
"
