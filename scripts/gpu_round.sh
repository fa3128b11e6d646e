#!/bin/bash
# One gpurun call: parity tests, smoke, bench, kernel trace.  Everything lands under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench.log 2>&1
REPO=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*.csv" | head; 
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
tail -n 3 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log
