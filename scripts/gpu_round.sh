#!/bin/bash
# One gpurun call: parity tests, smoke, bench, kernel trace, PMC pass.  Everything lands under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench.log 2>&1
timeout 600 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-graph > gpurun_out/bench_nograph.log 2>&1
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
# HBM traffic counters in their own pass (no other trace domains)
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc -o fetch -- python $REPO/bench.py --layers 4 --steps 4 --warmup 2 --prefill 512 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_fetch.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc -o write -- python $REPO/bench.py --layers 4 --steps 4 --warmup 2 --prefill 512 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_write.log 2>&1)
ls -la gpurun_out/pmc | head
find gpurun_out/pmc -name "*.csv" -size +20M -delete
tail -n 2 gpurun_out/pytest_gpu.log gpurun_out/smoke.log
for f in gpurun_out/bench.log gpurun_out/bench_nograph.log; do
grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'], 'TF', d.get('roofline',{}).get('achieved'))
"; done
