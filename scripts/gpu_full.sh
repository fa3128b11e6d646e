#!/bin/bash
# Full validation + evidence in one box: whole -m gpu suite, smoke, driver-style bench, profiles (kernel stats, one-step
# timeline, FETCH/WRITE_SIZE, MfmaUtil), launch-path profile, other BASELINE configs, the engine's phase timelines.
#   gpurun --timeout 3300 -- 'bash scripts/gpu_full.sh'   then here: python scripts/make_profiles.py r04 (+ copy the logs, profiles/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/final.log
timeout 3000 python -m pytest tests -m gpu -q -x -s > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5 | tee $LOG
grep -E "layer parity|8 layers, 4096|Mixtral-8x7B dims x 4|Mixtral-8x22B dims x 3|Nemo-12B dims x 4|max\|HIP|bit-exact" gpurun_out/pytest_gpu.log | cut -c1-500 | tee -a $LOG
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-500 | tee -a $LOG
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_steps20.json.log | cut -c1-400 | tee -a $LOG
python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_steps64.json.log | cut -c1-300 | tee -a $LOG
bash scripts/profile_round.sh pmc > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log | cut -c1-200
python bench.py --model nemo-12b --prefill 8192 --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_nemo12b.json.log | cut -c1-300 | tee -a $LOG
python bench.py --model mixtral-8x7b --steps 32 --warmup 4 2>&1 | tail -1 | tee gpurun_out/bench_mixtral8x7b.json.log | cut -c1-300 | tee -a $LOG
MI_ENGINE_VARIANT=2 python bench.py --model mixtral-8x7b --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_mixtral8x7b_shipped_engine_build.json.log | cut -c1-300 | tee -a $LOG
python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_8x22b_stage7.json.log | cut -c1-300 | tee -a $LOG
MI_DECODE_ENGINE=0 python bench.py --model mixtral-8x22b --layers 7 --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_8x22b_stage7_launch_path.json.log | cut -c1-300 | tee -a $LOG
MI_DECODE_ENGINE=0 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_launch_path.json.log | cut -c1-300 | tee -a $LOG
timeout 300 python scripts/engine_trace.py > gpurun_out/engine_trace.log 2>&1; tail -48 gpurun_out/engine_trace.log | tee -a $LOG
timeout 300 python scripts/engine_trace.py --model mixtral-8x7b --layers 8 > gpurun_out/engine_trace_8x7b.log 2>&1
timeout 300 python scripts/engine_trace.py --model mixtral-8x22b --layers 7 > gpurun_out/engine_trace_8x22b_stage7.log 2>&1
# round 6 additions: batch 2-4 lines, the batch-3 kernel table, the prefill attention probe + scaling
for b in 3 2 4; do python bench.py --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | tail -1 | tee gpurun_out/bench_batch$b.json.log | cut -c1-300 | tee -a $LOG; done
export TMPDIR=/tmp; REPO=$PWD; rm -rf gpurun_out/prof_b3
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_b3 -o decode -- python $REPO/bench.py --batch 3 --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof_b3.log 2>&1)
find gpurun_out/prof_b3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_batch3.csv
find gpurun_out/prof_b3 -name "*kernel_trace.csv" -delete
python scripts/attn_prefill_probe.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_prefill_probe.log | tee -a $LOG
python scripts/sampling_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sampling_probe.log | tee -a $LOG
