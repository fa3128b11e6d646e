#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash scripts/gpu_ab.sh ${1:-2}
