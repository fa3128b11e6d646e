#!/bin/bash
# round 4, call 3: remaining test fixes, PP sessions on real kernels, the stamp-site bisect, graph vs eager stepping,
# a phase timeline of the wide engine build on an 8x22B stage, and a seed search for the 8x22B depth test on THIS host
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/call3.log
: > $L
(cd tests && PYTHONPATH=../oracle timeout 900 python moe_depth_util.py 10 8x22b all 22 > ../gpurun_out/seed22b_box.log 2>&1) &
SEEDPID=$!
timeout 300 python -m pytest tests/test_gpu_sampling.py -q -x 2>&1 | tail -4 | tee -a $L
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x 2>&1 | tail -8 | tee -a $L
# graph replay vs eager stepping of the one-kernel step, interleaved
for rep in 1 2; do
  for mode in "" "--no-graph"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stepping [$mode]', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_us'], d['config']['decode_launch'][-20:])" | tee -a $L
  done
done
# stamp-site bisect (one rep each; main twice)
for n in main e_trace0 e_mask_loader e_mask_cons e_mask_cons_lo e_mask_cons_hi e_mask_even e_mask_odd main; do
  if [ $n == main ]; then unset MISTRAL_HIP_LIB; else export MISTRAL_HIP_LIB=$PWD/mistral-inference_amd/lib/variants/libmistral_hip_$n.so; fi
  timeout 300 python bench.py --steps 32 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sites $n', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_us'])" | tee -a $L
done
unset MISTRAL_HIP_LIB
timeout 300 python scripts/engine_trace.py --model mixtral-8x22b --layers 7 > gpurun_out/r04_engine_trace_8x22b_stage7_wide.txt 2>&1
tail -45 gpurun_out/r04_engine_trace_8x22b_stage7_wide.txt | tee -a $L
timeout 300 python scripts/engine_trace.py --model mixtral-8x7b --layers 8 > gpurun_out/r04_engine_trace_8x7b_8layers.txt 2>&1
tail -45 gpurun_out/r04_engine_trace_8x7b_8layers.txt | tee -a $L
wait $SEEDPID
cat gpurun_out/seed22b_box.log | tee -a $L
timeout 600 python -m pytest tests/test_gpu_depth.py -q -x -s -k "8x22b" 2>&1 | grep -E "passed|failed|rror|8x22B|assert" | cut -c1-600 | tee -a $L
