#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 1500 python bench.py --model mixtral-8x7b --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/bench_mixtral.log 2>&1
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1)
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o mixtral -- python $REPO/bench.py --model mixtral-8x7b --layers 8 --steps 8 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_mixtral.log 2>&1)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
tail -n 4 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench.log gpurun_out/bench_mixtral.log; do
tail -n 3 $f | grep -v '"metric"' | tail -n 2
grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'], 'TF', d.get('roofline',{}).get('achieved'))
"; done
