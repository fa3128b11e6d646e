#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 1500 python bench.py --model mixtral-8x7b --steps 48 --warmup 4 --no-cpu-baseline > gpurun_out/bench_mixtral.log 2>&1
rm -rf gpurun_out/pmc2
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc2 -o sq1 -- python $REPO/bench.py --layers 2 --steps 2 --warmup 2 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_sq1.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc2 -o sq2 -- python $REPO/bench.py --layers 2 --steps 2 --warmup 2 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_sq2.log 2>&1)
find gpurun_out/pmc2 -name "*.csv" -size +20M -delete
ls gpurun_out/pmc2
tail -n 3 gpurun_out/pytest_gpu.log; tail -n 3 gpurun_out/pmc_sq1.log | cut -c1-300; tail -n 3 gpurun_out/pmc_sq2.log | cut -c1-300
for f in gpurun_out/bench_mixtral.log; do
grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d['prefill']['tflops'], 'TF', d.get('roofline',{}).get('achieved'))
"; done
