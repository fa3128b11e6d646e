#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats + one-step timeline source, then FETCH_SIZE / WRITE_SIZE passes
# (counters in their own runs, kernel-trace only).  Everything lands under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof_bench.log 2>&1)
python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last complete decode steps: from one engine launch (or decode_prep on the launch path) to the next
    idx = [i for i, r in enumerate(rows) if "decode_engine_kernel" in r["Kernel_Name"]] or \
          [i for i, r in enumerate(rows) if "decode_prep" in r["Kernel_Name"]]
    idx = idx[-4:-1] if len(idx) >= 4 else idx
    if len(idx) >= 2:
        a, b = idx[0], idx[-1]
        t0 = int(rows[a]["Start_Timestamp"])
        with open("gpurun_out/one_step_timeline.csv", "w") as o:
            o.write("start_us,duration_us,kernel,grid_x,wg_x,vgpr,lds\n")
            for r in rows[a:b]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                o.write(f'{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},"{r["Kernel_Name"]}",{r["Grid_Size_X"]},{r["Workgroup_Size_X"]},{r.get("VGPR_Count", "")},{r.get("LDS_Block_Size", "")}\n')
PY
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
if [ "$1" == "pmc" ]; then
# the FULL headline configuration (32 layers, 4096-token prefill, decode at context 4096+): counters in their own passes
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc -o fetch -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $REPO/gpurun_out/pmc_fetch.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc -o write -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $REPO/gpurun_out/pmc_write.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc MfmaUtil VALUBusy LdsUtil LdsBankConflict --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc -o mfma -- python $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $REPO/gpurun_out/pmc_mfma.log 2>&1)
# the launch path (6 kernels per layer) for the per-kernel decode table
(cd /tmp && MI_DECODE_ENGINE=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_launch -o decode -- python $REPO/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof_launch.log 2>&1)
find gpurun_out/prof_launch -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_launch_path.csv
find gpurun_out/prof_launch -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/pmc -name "*.csv" -size +20M -delete
fi
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
head -30 gpurun_out/kernel_stats.csv | cut -c1-200
