#!/bin/bash
# Round 4: WHY is the decode engine 17 % slower without its never-taken trace-stamp branches (ENG_TRACE=0)?
# SQ / SQC counters of the shipped library and of the ENG_TRACE=0 variant (scripts/build_variants.py e_trace0), same box, the
# headline configuration, counters in their own passes (kernel-trace only, as the gpurun rules require).
#   bash scripts/engine_pmc.sh            -> gpurun_out/engine_pmc/{shipped,e_trace0}_<set>.csv + gpurun_out/engine_pmc/table.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/engine_pmc
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 -L > $OUT/counters_available.txt 2>&1)
export OUT
python - <<'PY' > $OUT/sets.txt
import re, os
txt = open(os.environ["OUT"] + "/counters_available.txt").read()
avail = set(re.findall(r"\b((?:SQ|SQC|GRBM)_[A-Z0-9_]+)\b", txt))
want = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU"],
    ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQ_IFETCH"],
    ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH_LEVEL"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]
for s in want:
    ok = [c for c in s if c in avail]
    if ok:
        print(" ".join(ok))
PY
cat $OUT/sets.txt
V=$REPO/mistral-inference_amd/lib/variants/libmistral_hip_e_trace0.so
for name in shipped e_trace0; do
  if [ $name == e_trace0 ]; then export MISTRAL_HIP_LIB=$V; else unset MISTRAL_HIP_LIB; fi
  # un-profiled timing of the same command first (a profiled pass clocks differently: never compare across)
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'ms_per_step', d['ms_per_step'], 'kernel_us', d['roofline']['avg_launch_us'])" | tee -a $OUT/table.txt
  i=0
  while read -r set; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw_${name}_$i -o pmc -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $OUT/log_${name}_$i.txt 2>&1) || echo "set $i failed for $name" | tee -a $OUT/table.txt
  done < $OUT/sets.txt
done
unset MISTRAL_HIP_LIB
python - <<'PY' | tee -a $OUT/table.txt
import csv, glob, os, collections
out = os.environ["OUT"]
tab = collections.OrderedDict()
for name in ("shipped", "e_trace0"):
    for d in sorted(glob.glob(f"{out}/raw_{name}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "decode_engine_kernel" in r.get("Kernel_Name", ""):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, v in acc.items():
                v = v[-4:]                      # the last launches (steady state)
                tab.setdefault(c, {})[name] = sum(v) / len(v)
print(f"{'counter (per engine launch, mean of the last launches)':58s} {'shipped':>16s} {'ENG_TRACE=0':>16s} {'ratio':>8s}")
for c, v in tab.items():
    a, b = v.get("shipped"), v.get("e_trace0")
    if a is not None and b is not None:
        print(f"{c:58s} {a:16.0f} {b:16.0f} {b / a if a else float('nan'):8.3f}")
PY
find $OUT -name "*.csv" -size +8M -delete
