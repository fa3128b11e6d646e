#!/bin/bash
# end-of-round check: full GPU suite, smoke, driver-style bench, a 2-rank pipeline bench smoke run (gloo transport on one GPU)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/final.log
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5 | tee $LOG
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $LOG
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_r02_steps20.json.log | cut -c1-300 | tee -a $LOG
MI_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 --layers 8 --prefill 1024 2>&1 | tail -2 | cut -c1-700 | tee -a $LOG
