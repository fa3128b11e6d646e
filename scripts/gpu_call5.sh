#!/bin/bash
# round 4, call 5+: the MoE engine builds (router on the holder waves, two-pass ratio-6 attention): parity + speed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/call5.log
: > $L
timeout 300 python -m pytest tests/test_gpu_sampling.py -q 2>&1 | tail -3 | tee -a $L
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "moe or wide or gqa6" 2>&1 | tail -4 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_depth.py -q -x -s -k "mixtral" 2>&1 | grep -E "passed|failed|rror|Mixtral|assert" | cut -c1-300 | tee -a $L
for v in 0 2 0 2; do
  MI_ENGINE_VARIANT=$v timeout 500 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_mixtral8x7b_variant$v.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x7b variant $v', d['value'], d['ms_per_step'], d['hbm_roofline_step']['frac'], d['prefill']['tokens_per_s'])" | tee -a $L
done
MI_DECODE_ENGINE=0 timeout 500 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x7b launch path', d['value'], d['ms_per_step'], d['hbm_roofline_step']['frac'])" | tee -a $L
for e in 1 0 1 0; do
  MI_DECODE_ENGINE=$e timeout 400 python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r04_bench_8x22b_stage7_engine$e.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x22b stage engine=$e', d['value'], d['ms_per_step'], d['hbm_roofline_step']['frac'])" | tee -a $L
done
timeout 300 python scripts/engine_trace.py --model mixtral-8x7b --layers 8 > gpurun_out/r04_engine_trace_8x7b_8layers_moe_build.txt 2>&1
sed -n 1,50p gpurun_out/r04_engine_trace_8x7b_8layers_moe_build.txt | tee -a $L
timeout 300 python scripts/engine_trace.py --model mixtral-8x22b --layers 7 > gpurun_out/r04_engine_trace_8x22b_stage7_wide_final.txt 2>&1
sed -n 1,30p gpurun_out/r04_engine_trace_8x22b_stage7_wide_final.txt | tee -a $L
