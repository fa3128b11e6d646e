#!/bin/bash
# Prefill attention A/B: every variant library named on the command line (scripts/build_variants.py a_*), then the main build, through
# scripts/attn_prefill_probe.py on one box.   gpurun -- 'bash scripts/gpu_attn_ab.sh a_fast0 a_prio2'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=mistral-inference_amd/lib/variants
for n in main "$@" main; do
  echo "== $n"
  if [ $n = main ]; then python scripts/attn_prefill_probe.py 5 2>&1 | grep -v amdgpu.ids; else MISTRAL_HIP_LIB=$V/libmistral_hip_$n.so python scripts/attn_prefill_probe.py 5 2>&1 | grep -v amdgpu.ids; fi
done
