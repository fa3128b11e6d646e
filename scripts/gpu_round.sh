#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --timeout 900 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/bench.log 2>&1
grep -h '"metric"' gpurun_out/bench.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', 'step frac', d['hbm_roofline_step']['frac'], 'prefill', d['prefill']['tokens_per_s'], d.get('roofline',{}).get('achieved'))
"
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv,re
rows = list(csv.DictReader(open("gpurun_out/prof/decode_kernel_stats.csv")))
for r in rows[:16]:
    n = re.sub(r"\(anonymous namespace\)::","",r["Name"])[:60]
    if "at::native" in n: continue
    print(f"{n:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
