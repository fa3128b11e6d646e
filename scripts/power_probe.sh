#!/bin/bash
# Board power and clocks (rocm-smi, every ~0.5 s) while bench.py's decode loop runs 4000 steps:
#   gpurun --timeout 60 -- 'bash scripts/power_probe.sh'
# Round 3: 1166-1179 W at sclk 2.39 GHz, mclk 2.0 GHz - the decode engine is NOT at the 1400 W limit (the prefill GEMMs are:
# scripts/prefill_probe.py --power).  The 4000-step run itself: 2.6276 ms per step = 380.6 tokens/s = 70.2 % of 8 TB/s.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 100 python bench.py --steps 4000 --warmup 8 --no-cpu-baseline --no-extras > /tmp/b.log 2>&1 &
BP=$!
for i in $(seq 1 40); do
  timeout 5 rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import json,sys
try:
    c=json.load(sys.stdin)['card0']; print('$i', c.get('Current Socket Graphics Package Power (W)'), c.get('sclk clock speed:'), c.get('mclk clock speed:'), c.get('fclk clock speed:'))
except Exception as e: print('$i err', e)"
  sleep 0.4
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -1 /tmp/b.log | cut -c1-200
