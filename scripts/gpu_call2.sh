#!/bin/bash
# round 4, call 2: wide engine build (bit equality + speed on the Nemo / 8x22B-stage shapes), PP sessions on real kernels,
# the sampling and 8x22B depth fixes, and the ENG_TRACE PMC comparison
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/call2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "wide" 2>&1 | tail -12 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -8 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_depth.py -q -x -s -k "8x22b" 2>&1 | grep -E "passed|failed|rror|8x22B|assert" | cut -c1-600 | tee -a $L
for v in 1 0; do
  MI_DECODE_ENGINE=$v timeout 400 python bench.py --model nemo-12b --prefill 8192 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_nemo12b_engine$v.json.log | cut -c1-700 | tee -a $L
  MI_DECODE_ENGINE=$v timeout 400 python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_8x22b_stage7_engine$v.json.log | cut -c1-700 | tee -a $L
done
timeout 900 bash scripts/engine_pmc.sh > gpurun_out/engine_pmc.log 2>&1
tail -40 gpurun_out/engine_pmc/table.txt | tee -a $L
