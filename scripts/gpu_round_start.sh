#!/bin/bash
# First GPU call of a round: everything that re-establishes the state of the tree on hardware, in one box
# (~12 GPU-minutes).  Before calling, build the A/B variant here:
#   (cd mistral-inference_amd && python -c "import build_native as b; b.build(); \
#      b.build(extra_flags=('-DENG_HOLDERS=0',), obj_dir='/tmp/obj_h0', lib='lib/variants/libmistral_hip_holders0.so')")
#   gpurun --timeout 1200 -- 'bash scripts/gpu_round_start.sh'
# Then: python scripts/make_profiles.py rNN   (copies the summaries into profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash scripts/gpu_final.sh                       # full -m gpu suite, smoke, driver-style bench, 2-rank pipeline smoke run
[ -f mistral-inference_amd/lib/variants/libmistral_hip_holders0.so ] && bash scripts/gpu_holders.sh   # holder waves A/B + trace
bash scripts/profile_round.sh pmc > gpurun_out/profile_round.log 2>&1    # rocprofv3 kernel stats + FETCH/WRITE_SIZE passes
tail -5 gpurun_out/profile_round.log
