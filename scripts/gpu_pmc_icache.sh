#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc_icache
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|SQC_INST|IFETCH" | head -20 > gpurun_out/pmc_icache/counters.txt
cd /tmp
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_icache/$tag -o run -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_icache/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_icache/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "decode_engine" in k or "gemv_kernel" in k:
            a = agg[(k, r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items()):
        print(f"{k:62s} {c:22s} n={n:4d} avg={v / n:14.1f}")
PY
