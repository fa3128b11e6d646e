#!/bin/bash
# Slim end-of-round validation in one box (the other BASELINE configurations were measured earlier in the round on the same
# kernels: scripts/gpu_full.sh): whole -m gpu suite, smoke, driver-style bench, kernel stats + one-step timeline, generic probe.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_final.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
exec < /dev/null
mkdir -p gpurun_out
LOG=gpurun_out/final.log
: > $LOG
timeout 1000 python -m pytest tests -m gpu -q -x -s > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5 | tee -a $LOG
grep -E "layer parity|8 layers, 4096|Mixtral-8x7B dims x 4|Mixtral-8x22B dims x 3|Nemo-12B dims x 4|max\|HIP|bit-exact" gpurun_out/pytest_gpu.log | cut -c1-400 >> $LOG
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-500 | tee -a $LOG
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_steps20.json.log | cut -c1-600 | tee -a $LOG
timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_steps64.json.log | cut -c1-300 | tee -a $LOG
[ -n "$FINAL_SKIP_PROFILES" ] || { timeout 900 bash scripts/profile_round.sh pmc > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log | cut -c1-200; }
timeout 200 python scripts/generic_probe.py 4 2048 > gpurun_out/generic_probe.log 2>&1; tail -1 gpurun_out/generic_probe.log | cut -c1-700 | tee -a $LOG
# the engine builds side by side on THIS box (frozen default object, the routed `next` build, its twin with stamp sites for the
# timeline) - needs lib/variants/libmistral_hip_slots.so (python scripts/build_variants.py engine_slots)
[ -f mistral-inference_amd/lib/variants/libmistral_hip_slots.so ] && MISTRAL_HIP_LIB=$PWD/mistral-inference_amd/lib/variants/libmistral_hip_slots.so \
  timeout 400 python scripts/engine_ab.py --steps 200 --reps 3 --only ns_trace0,nst_hid,ce_hid,ns_ce_hid --trace-names ns_ce_hid > gpurun_out/engine_ab.stdout 2>&1
grep -A8 "^entry" gpurun_out/engine_ab.stdout | cut -c1-120 | tee -a $LOG
# the N > 1 line's shape on this 1-GPU box: two ranks share the GPU over gloo (launch path: two engines cannot be resident together)
MI_DIST_BACKEND=gloo MI_DECODE_ENGINE=0 MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29391 \
  bench.py --gpus 2 --steps 8 --warmup 2 --layers 8 --prefill 512 --mixtral-layers 2 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_2ranks_gloo_one_gpu.json.log
cut -c1-400 gpurun_out/bench_2ranks_gloo_one_gpu.json.log | tee -a $LOG
for b in 3; do timeout 300 python bench.py --batch $b --steps 64 --warmup 8 2>&1 | tail -1 > gpurun_out/bench_batch$b.json.log; cut -c1-300 gpurun_out/bench_batch$b.json.log | tee -a $LOG; done
# the other BASELINE configurations as lines of their own (they are also sub-objects of the steps-20 line above)
timeout 300 python bench.py --model mixtral-8x22b --layers 7 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_8x22b_stage7.json.log; cut -c1-200 gpurun_out/bench_8x22b_stage7.json.log | tee -a $LOG
timeout 400 python bench.py --model mixtral-8x7b --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_mixtral8x7b.json.log; cut -c1-200 gpurun_out/bench_mixtral8x7b.json.log | tee -a $LOG
timeout 400 python bench.py --model nemo-12b --prefill 8192 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_nemo12b.json.log; cut -c1-200 gpurun_out/bench_nemo12b.json.log | tee -a $LOG
MI_DECODE_ENGINE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_launch_path.json.log; cut -c1-200 gpurun_out/bench_launch_path.json.log | tee -a $LOG
