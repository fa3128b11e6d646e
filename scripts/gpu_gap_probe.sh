#!/bin/bash
# Where does (step time - engine kernel time) go?  rocprofv3 kernel traces of the greedy loop (graph and eager) and of the
# forward()+argmax loop: per-kernel durations and the idle gaps between consecutive kernels of the timed steps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
probe() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf gpurun_out/gap_$label
  (cd /tmp && env "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/gap_$label -o t -- python $REPO/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras "$@" > $REPO/gpurun_out/gap_$label.log 2>&1)
  python - "$label" <<'PY'
import csv, glob, sys, json
label = sys.argv[1]
f = glob.glob(f"gpurun_out/gap_{label}/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
eng = [i for i, r in enumerate(rows) if "decode_engine_kernel" in r["Kernel_Name"]]
# the timed steps of the session/loop: take the engine launches 8..28 (after warm-up), list what runs between consecutive ones
import statistics as st
durs, gaps, between = [], [], {}
for a, b in zip(eng[8:28], eng[9:29]):
    ra, rb = rows[a], rows[b]
    durs.append((int(ra["End_Timestamp"]) - int(ra["Start_Timestamp"])) / 1e3)
    gaps.append((int(rb["Start_Timestamp"]) - int(ra["End_Timestamp"])) / 1e3)
    for r in rows[a + 1:b]:
        n = r["Kernel_Name"][:60]
        between.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
line = open(f"gpurun_out/gap_{label}.log").read().strip().splitlines()[-1]
try:
    ms = json.loads(line)["ms_per_step"]
except Exception:
    ms = None
print(f"{label}: engine kernel us median {st.median(durs):.1f} (min {min(durs):.1f} max {max(durs):.1f}); end->next start gap us median {st.median(gaps):.1f} (min {min(gaps):.1f} max {max(gaps):.1f}); bench ms/step under the profiler {ms}")
for n, v in between.items():
    print(f"    between engine launches: {len(v) / len(durs):.1f} x {n}  {st.median(v):.1f} us")
PY
  find gpurun_out/gap_$label -name "*.csv" -size +5M -delete
}
probe greedy_graph MI_ENGINE_BALANCE=0 MI_GRAPH_STEPS=1 --
probe greedy_graph8 MI_ENGINE_BALANCE=0 MI_GRAPH_STEPS=8 --
probe greedy_eager MI_ENGINE_BALANCE=0 -- --no-graph
probe forward_graph MI_ENGINE_BALANCE=0 -- --loop forward
