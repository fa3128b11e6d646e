#!/usr/bin/env python3
"""Build A/B variants of libmistral_hip.so for scripts/gpu_ab.sh (compile-time switches of the decode engine).

    python scripts/build_variants.py                 # the three round-3 experiments, each switched ON, next to the shipped build
    gpurun --timeout 900 -- 'bash scripts/gpu_ab.sh 2'

A variant is `name: (extra hipcc flags...)`; objects go to /tmp/obj_<name>, the library to
mistral-inference_amd/lib/variants/libmistral_hip_<name>.so (git-ignored; travels to the GPU box with gpurun)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd")
sys.path.insert(0, ROOT)
import build_native as b  # noqa: E402

VARIANTS = {
    "asm_dma1": ("-DENG_ASM_DMA=1",),             # the loader's DMA through the builtin (hipcc then decides where vmcnt waits go)
    "trace0": ("-DENG_TRACE=0",),                 # no stamp sites: measured 17 % SLOWER (they pin the compiler's scheduling)
    "trace2": ("-DENG_TRACE=2",),                 # stamp sites replaced by bare compiler / scheduling barriers
    "all4_0": ("-DENG_ALL4=0",),                  # the generic-group path compiled into every instantiation (round-2 form)
    "cbar_flags": ("-DENG_CBAR_FLAGS=1",),        # consumer barrier on per-wave flag words (measured +10..20 us per step)
    "sparse_poll": ("-DENG_SPARSE_POLL=1",),      # re-poll only the granules that were missing (+25 us)
    "lean_barriers": ("-DENG_LEAN_BARRIERS=1",),  # attn sweep starts while wave 0 still merges (+45 us)
    "holders0": ("-DENG_HOLDERS=0",),             # no holder waves (5-wave workgroups)
    # compiler-flag lottery (the kernel's speed is a chaotic function of its code: profiles/EXPERIMENTS.md)
    # (the shipped engine build uses -amdgpu-sched-strategy=max-memory-clause: build_native.PER_FILE_FLAGS; set
    #  MI_ENGINE_FLAGS="" in the environment of this script to get the default strategy as the baseline of such an A/B)
    "O2": ("-O2",),
    "maxilp": ("-mllvm", "-amdgpu-sched-strategy=max-ilp"),
    "nopostmisched": ("-mllvm", "-enable-post-misched=0"),
    "noslp": ("-fno-slp-vectorize",),
}

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "lib", "variants"), exist_ok=True)
    b.build(verbose=False)
    for name, flags in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(name, b.build(verbose=False, extra_flags=flags, obj_dir=f"/tmp/obj_{name}",
                            lib=os.path.join(ROOT, "lib", "variants", f"libmistral_hip_{name}.so")), flush=True)
