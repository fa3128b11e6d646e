#!/usr/bin/env python3
"""Build libmistral_hip.so (gfx950) in-tree: mistral-inference_amd/lib/libmistral_hip.so.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libmistral_hip.so")
SOURCES = ["api.hip", "gemv.hip", "gemm.hip", "gemm256.hip", "attn_decode.hip", "attn_prefill.hip", "elementwise.hip",
           "decode_engine.hip", "sampling.hip", "rccl_api.hip", "generic.hip"]
# (this file is a dependency of every object: a change of flags must rebuild them - round 6: decode_engine_next.o was once shipped
# without a flag that had just been added here)
HEADERS = [os.path.abspath(__file__), os.path.join(CSRC, "common.cuh"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "gemv_core.cuh"),
           os.path.join(CSRC, "attn_decode_core.cuh"), os.path.join(HERE, "..", "scripts", "probes", "gemm256_experiments.inc"),
           os.path.join(HERE, "..", "include", "mistral_hip.h"), os.path.join(HERE, "..", "include", "mistral_hip_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-file flags.  The persistent decode engine is built with the max-memory-clause scheduling strategy: measured 1.3-1.5 % faster
# on a good lease and 3 % on a slow one than the default strategy (same-box A/B of five flag variants, profiles/EXPERIMENTS.md) -
# the kernel's speed is a chaotic function of its code, so this is an observation, not a principle.  MI_ENGINE_FLAGS overrides.
PER_FILE_FLAGS = {"decode_engine.hip": os.environ.get("MI_ENGINE_FLAGS", "-mllvm -amdgpu-sched-strategy=max-memory-clause").split()}


# Second compile of the engine source under other entry-point names (csrc/decode_engine.hip, ENG_WIDE): the shapes the shipped
# instantiations decline, without touching one instruction of the shipped kernels.
# round 5: the headline shape's engine (dense GQA-4, rows of 4-piece groups) - the winner of the same-box A/B of
# scripts/build_variants.py engine_slots / scripts/engine_ab.py (profiles/EXPERIMENTS.md round 5)
# (abort word read every 1024th spin; consumers at s_setprio 1; holders fetch from the K/V stage on; the loader's weight DMAs from
#  inline asm in the SGPR-base form - which is what makes the build WITHOUT the debug stamp sites as fast as the one with them)
# round 6: + the loader is not stopped during the hid sweep (ENG_NOSTOP=32: the ring is empty there; +0.1..0.7 % on six boxes)
ENGINE_NEXT_FLAGS = ["-DENG_SUFFIX=_next", "-DENG_HEADLINE_ONLY=1", "-DENG_ABORT_RARE=1", "-DENG_CONS_PRIO=1", "-DENG_HOLD_STAGE=2",
                     "-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32"]
# round 6: Mistral-Nemo dims (dim 5120: rows of 10 pieces, contiguous units) on the 8-fill ring with every DMA from inline asm
ENGINE_NEMO_FLAGS = ["-DENG_SUFFIX=_nemo", "-DENG_HEADLINE_ONLY=2", "-DENG_WIDE=2", "-DENG_ABORT_RARE=1", "-DENG_CONS_PRIO=1",
                     "-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32", "-DENG_CLEAN_ENTRY=1"]  # (without the clean entry: 52 leaked
# `vmcnt(4)` guards inside this build's loader loops - scripts/engine_loader_waits.py)
VARIANT_OBJECTS = {"decode_engine_next.o": ("decode_engine.hip", ENGINE_NEXT_FLAGS),
                   "decode_engine_nemo.o": ("decode_engine.hip", ENGINE_NEMO_FLAGS),
                   # (the wide build also takes the two round-5 switches: +0.7 % on the 8x22B stage; the 8-fill MoE build does NOT -
                   #  the same two switches make Mixtral-8x7B 4.5 % slower, profiles/EXPERIMENTS.md round 5; and the q|k|v holder
                   #  waves, one unit each at rows of 12 pieces: +1.2 % on the 8x22B stage)
                   "decode_engine_wide.o": ("decode_engine.hip", ["-DENG_WIDE=1", "-DENG_ABORT_RARE=1", "-DENG_CONS_PRIO=1", "-DENG_QKV_HOLD=2"]),   # 7-fill ring, GQA 4 / 6, contiguous units
                   # 8-fill ring, MoE GQA 4 (Mixtral-8x7B); round 5: its idle holder waves keep six q|k|v units of the NEXT layer,
                   # fetched during the router bubble (ENG_QKV_HOLD = 2: +1.9 % on one box, +-0 on another)
                   "decode_engine_moe.o": ("decode_engine.hip", ["-DENG_WIDE=2", "-DENG_QKV_HOLD=2"]),
                   "gemm256_f16.o": ("gemm256.hip", ["-DG256_F16=1"]),                # the 256-tile GEMM on fp16 payloads (generic path)
                   "attn_prefill_f16.o": ("attn_prefill.hip", ["-DATTN_F16=1"]),      # the MFMA prefill attention on fp16 payloads
                   "gemv_f16.o": ("gemv.hip", ["-DGEMV_F16=1"])}                      # the weight-streaming GEMV kernels on fp16 payloads


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    missing = [d for d in deps if not os.path.exists(d)]
    if missing:  # a header that was moved or renamed must not silently stop triggering rebuilds
        raise RuntimeError(f"build_native: dependency listed but not found: {missing}")
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = True, extra_flags=(), obj_dir: str = OBJ, lib: str = LIB) -> str:
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, *PER_FILE_FLAGS.get(src, []), *extra_flags, "-c", s, "-o", o])
    for obj, (src, defs) in VARIANT_OBJECTS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, obj)
        if _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, *PER_FILE_FLAGS.get(src, []), *extra_flags, *defs, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES] + [os.path.join(obj_dir, o) for o in VARIANT_OBJECTS]
    if jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", lib])
    return lib


if __name__ == "__main__":
    print(build(verbose="-q" not in sys.argv))
