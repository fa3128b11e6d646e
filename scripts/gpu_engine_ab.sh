#!/bin/bash
# One box: the in-process A/B of the engine slots library (scripts/engine_ab.py) + the parity tests of the routed `next` build.
#   python scripts/build_variants.py engine_slots && gpurun --timeout 900 -- 'bash scripts/gpu_engine_ab.sh [engine_ab args]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
exec < /dev/null
mkdir -p gpurun_out
MISTRAL_HIP_LIB=$PWD/mistral-inference_amd/lib/variants/libmistral_hip_slots.so timeout 600 python scripts/engine_ab.py "$@" > gpurun_out/engine_ab.stdout 2>&1
tail -5 gpurun_out/engine_ab.stdout | cut -c1-300
head -40 gpurun_out/engine_ab.log | cut -c1-200
[ -n "$AB_SKIP_TESTS" ] || timeout 500 python -m pytest tests/test_gpu_engine.py -q -x -k "next_engine or full_size" 2>&1 | tail -5 | tee gpurun_out/engine_next_tests.log
