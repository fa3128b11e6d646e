#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/profile_round.sh (under gpurun_out/) into the committed summaries in profiles/.

    bash scripts/profile_round.sh pmc      # on the GPU box (via gpurun)
    python scripts/make_profiles.py r01    # here
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

shutil.copy(os.path.join(OUT, "kernel_stats.csv"), os.path.join(PROF, f"{tag}_decode_bench_kernel_stats.csv"))
shutil.copy(os.path.join(OUT, "one_step_timeline.csv"), os.path.join(PROF, f"{tag}_decode_one_step_timeline.csv"))
if os.path.exists(os.path.join(OUT, "kernel_stats_launch_path.csv")):
    shutil.copy(os.path.join(OUT, "kernel_stats_launch_path.csv"), os.path.join(PROF, f"{tag}_decode_launch_path_kernel_stats.csv"))

rows = []
per = {}
for counter, stem in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    path = os.path.join(OUT, "pmc", f"{stem}_counter_collection.csv")
    if not os.path.exists(path):
        continue
    # one row per (dispatch, counter instance): sum the instances of a dispatch, then average over dispatches
    disp = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        disp[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    byk = collections.defaultdict(list)
    for d, v in disp.items():
        byk[name[d]].append(v)
    for k, vs in sorted(byk.items()):
        if "at::native" in k or "rocclr" in k:
            continue
        short = k.replace("(anonymous namespace)::", "")
        rows.append((short, counter, len(vs), sum(vs) / len(vs), min(vs), max(vs)))
        per.setdefault(short, {})[counter] = sum(vs) / len(vs)
with open(os.path.join(PROF, f"{tag}_pmc_fetch_write_size.csv"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); bench.py --steps 4 --warmup 2 --no-graph: the FULL headline config (32 layers, 4096-token prefill, decode at context 4096+)\n")
    f.write("# values in KiB as reported; on gfx950 FETCH_SIZE counts 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2 * FETCH_SIZE * 1024\n")
    f.write("kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
    for k, c, n, m, lo, hi in rows:
        f.write(f'"{k}",{c},{n},{m:.1f},{lo:.1f},{hi:.1f}\n')
dom = [k for k in per if "decode_engine_kernel" in k] or [k for k in per if "gemv_kernel<1, 2, 2" in k]
if dom:
    k = dom[0]
    fetch, write = per[k].get("FETCH_SIZE", 0.0), per[k].get("WRITE_SIZE", 0.0)
    json.dump({"kernel": k, "fetch_size_KiB": fetch, "write_size_KiB": write,
               "hbm_bytes_per_launch": int(round(2 * fetch * 1024 + write * 1024)),
               "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-stream halving, MI355X_MICROARCH.md section HBM); WRITE_SIZE uncorrected",
               "source": f"profiles/{tag}_pmc_fetch_write_size.csv"},
              open(os.path.join(PROF, "pmc_dominant_kernel.json"), "w"), indent=1)
    print("dominant kernel HBM bytes per launch:", int(round(2 * fetch * 1024 + write * 1024)))
mf = os.path.join(OUT, "pmc", "mfma_counter_collection.csv")
if os.path.exists(mf):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(mf)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "at::native" in k or "rocclr" in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(PROF, f"{tag}_pmc_mfma_util.csv"), "w") as f:
        f.write("# rocprofv3 --pmc MfmaUtil VALUBusy LdsUtil LdsBankConflict --kernel-trace (own pass); bench.py --steps 2 --warmup 2 --no-graph (32 layers, 4096-token prefill)\n")
        f.write("# derived metrics, mean over dispatches (gfx950 falls back to the gfx94x formulas, MI355X_MICROARCH.md 'rocprofv3 PMC slots')\n")
        f.write("kernel,dispatches,MfmaUtil_pct,VALUBusy_pct,LdsUtil_pct,LdsBankConflict_per_access\n")
        for k, d in sorted(acc.items()):
            m = lambda c: sum(d[c]) / len(d[c]) if d.get(c) else 0.0
            f.write(f'"{k}",{len(next(iter(d.values())))},{m("MfmaUtil"):.1f},{m("VALUBusy"):.1f},{m("LdsUtil"):.1f},{m("LdsBankConflict"):.3f}\n')
print("wrote", len(rows), "PMC rows")
