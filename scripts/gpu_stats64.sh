#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command's decode loop (--steps 64 --warmup 8): the dominant kernel's average
# with the steady-state launches in the majority (profile_round.sh profiles --steps 16 --warmup 2, where the launches behind the
# run's idle periods - clock ramp, scripts/first_step_probe.py - are a quarter of all launches).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/prof64
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof64 -o decode -- python $REPO/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof64_bench.log 2>&1)
find gpurun_out/prof64 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_steps64.csv
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof64/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "decode_engine_kernel" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
n = len(d)
print(f"decode_engine_kernel: {n} launches, mean {sum(d)/n:.1f} us, median {d[n//2]:.1f}, min {d[0]:.1f}, p90 {d[int(n*0.9)]:.1f}, max {d[-1]:.1f}; launches above median + 2 %: {sum(x > d[n//2]*1.02 for x in d)}")
PY
find gpurun_out/prof64 -name "*kernel_trace.csv" -size +20M -delete
tail -1 gpurun_out/prof64_bench.log | cut -c1-200
head -3 gpurun_out/kernel_stats_steps64.csv | cut -c1-200
