cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/prof_nemo
for v in 1 0 1 0; do MI_GEMV_NW=$v timeout 300 python bench.py --model nemo-12b --prefill 8192 --steps 32 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MI_GEMV_NW=$v nemo', d['value'], d['ms_per_step'], d['hbm_roofline_step']['frac'])"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_nemo -o decode -- python $REPO/bench.py --model nemo-12b --prefill 8192 --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof_nemo.log 2>&1)
find gpurun_out/prof_nemo -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_nemo.csv
find gpurun_out/prof_nemo -name "*kernel_trace.csv" -size +20M -delete
grep "gemv_kernel\|attn_decode\|greedy\|prep" gpurun_out/kernel_stats_nemo.csv | cut -d, -f1-4 | cut -c1-170
