#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the launch-path decode kernels at the FULL headline configuration (MI_DECODE_ENGINE=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out/pmc_launch
export TMPDIR=/tmp
(cd /tmp && MI_DECODE_ENGINE=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_launch -o fetch -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_launch/fetch.log 2>&1)
(cd /tmp && MI_DECODE_ENGINE=0 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_launch -o write -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $REPO/gpurun_out/pmc_launch/write.log 2>&1)
python - <<'PY'
import collections, csv
rows = []
for counter, stem in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    disp, name = collections.defaultdict(float), {}
    for r in csv.DictReader(open(f"gpurun_out/pmc_launch/{stem}_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            disp[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    byk = collections.defaultdict(list)
    for d, v in disp.items():
        byk[name[d]].append(v)
    for k, vs in sorted(byk.items()):
        if "gemv_kernel" in k or "attn_decode" in k:
            rows.append((k.replace("(anonymous namespace)::", ""), counter, len(vs), sum(vs) / len(vs), min(vs), max(vs)))
with open("gpurun_out/pmc_launch_path_decode.csv", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); MI_DECODE_ENGINE=0 bench.py --steps 4 --warmup 2 --no-graph: launch-path decode kernels at context 4096+ (32 layers)\n")
    f.write("# values in KiB as reported; gfx950: hbm_read_bytes = 2 * FETCH_SIZE * 1024 for wide coalesced streams (MI355X_MICROARCH.md, HBM)\n")
    f.write("kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
    for k, c, n, m, lo, hi in rows:
        f.write(f'"{k}",{c},{n},{m:.1f},{lo:.1f},{hi:.1f}\n')
print(open("gpurun_out/pmc_launch_path_decode.csv").read())
PY
