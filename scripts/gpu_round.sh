#!/bin/bash
# One gpurun call: parity tests, smoke, bench, kernel trace.  Everything lands under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
MI_ATTN_TWO_PASS=1 timeout 900 python -m pytest tests/test_gpu_ops.py -k attn_decode -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_twopass.log 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench.log 2>&1
MI_ATTN_TWO_PASS=1 timeout 600 python bench.py --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/bench_twopass.log 2>&1
MI_GEMV_SINGLE_BELOW=0 timeout 600 python bench.py --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/bench_pairs.log 2>&1
MI_ATTN_SPLIT_SLOTS=64 timeout 600 python bench.py --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/bench_split64.log 2>&1
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
tail -n 2 gpurun_out/pytest_gpu.log gpurun_out/pytest_twopass.log gpurun_out/smoke.log
for f in gpurun_out/bench.log gpurun_out/bench_twopass.log gpurun_out/bench_pairs.log gpurun_out/bench_split64.log; do
grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'], 'TF', d.get('roofline',{}).get('achieved'))
"; done
