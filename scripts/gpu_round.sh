#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider --timeout 400 > gpurun_out/pytest_pp.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_pp.log
MI_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 16 --warmup 4 > gpurun_out/bench_pp2_gloo.log 2>&1
echo "pp2 exit $?" >> gpurun_out/bench_pp2_gloo.log
tail -n 6 gpurun_out/pytest_pp.log; tail -n 4 gpurun_out/bench_pp2_gloo.log | cut -c1-600
