#!/bin/bash
# One more sample of the headline numbers on whatever box the pool hands out (boxes differ by up to 12 %), plus the soak test.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "soak" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/sample_steps20.json.log; cut -c1-200 gpurun_out/sample_steps20.json.log
python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/sample_steps64.json.log; cut -c1-200 gpurun_out/sample_steps64.json.log
MI_DECODE_ENGINE=0 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/sample_launch_path.json.log; cut -c1-200 gpurun_out/sample_launch_path.json.log
python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/sample_mixtral.json.log; cut -c1-200 gpurun_out/sample_mixtral.json.log
