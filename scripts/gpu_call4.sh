#!/bin/bash
# round 4, call 4: MoE on the wide engine build (batched router, two-pass ratio-6 attention): parity + speed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/call4.log
: > $L
timeout 300 python -m pytest tests/test_gpu_sampling.py -q 2>&1 | tail -4 | tee -a $L
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q 2>&1 | tail -6 | tee -a $L
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_greedy.py -q -x 2>&1 | tail -6 | tee -a $L
timeout 900 python -m pytest tests/test_gpu_depth.py -q -x -s -k "mixtral" 2>&1 | grep -E "passed|failed|rror|Mixtral|assert" | cut -c1-400 | tee -a $L
for v in 0 2; do
  MI_ENGINE_VARIANT=$v timeout 500 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_mixtral8x7b_variant$v.json.log | cut -c1-420 | tee -a $L
done
timeout 400 python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_8x22b_stage7_wide_engine_v2.json.log | cut -c1-420 | tee -a $L
MI_DECODE_ENGINE=0 timeout 400 python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee -a $L
timeout 300 python scripts/engine_trace.py --model mixtral-8x7b --layers 8 > gpurun_out/r04_engine_trace_8x7b_8layers_wide.txt 2>&1
sed -n 1,30p gpurun_out/r04_engine_trace_8x7b_8layers_wide.txt | tee -a $L
timeout 300 python scripts/engine_trace.py --model mixtral-8x22b --layers 7 > gpurun_out/r04_engine_trace_8x22b_stage7_wide_v2.txt 2>&1
sed -n 1,30p gpurun_out/r04_engine_trace_8x22b_stage7_wide_v2.txt | tee -a $L
