#!/usr/bin/env python3
"""Byte-compile the UNMODIFIED reference package into oracle/_ref/ (test / bench infrastructure, never product code).

Why: `bench.py`'s `cpu_baseline` leg wants the reference's own CPU path timed on the GPU box's host cores
("kind": "reference"), and /root/reference does not exist there.  The reference is pure Python, so the analogue of
"compile the C reference from its sources where they lie into oracle/_ref/" is `py_compile`: every module of
/root/reference/src/mistral_inference is compiled FROM THE SOURCE WHERE IT LIES into a sourceless
oracle/_ref/mistral_inference/<module>.pyc.  No reference source text is copied into the repository; oracle/_ref/ is
git-ignored (it never enters the history) but not gpurun-ignored (it travels to the GPU box with the working tree, like the
built libmistral_hip.so).  Same interpreter on both sides (this image's /usr/bin/python3), so the bytecode loads there.

    python oracle/build_ref.py            # no-op (exit 0) where /root/reference is absent
Consumers: oracle/time_reference.py (MISTRAL_REFERENCE_SRC falls back to oracle/_ref), through oracle/shim as always.
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")
DST = os.path.join(HERE, "_ref")


def build(verbose: bool = True) -> bool:
    pkg = os.path.join(SRC, "mistral_inference")
    if not os.path.isdir(pkg):
        if verbose:
            print(f"oracle/build_ref.py: {pkg} not found - nothing to do (the GPU box uses the prebuilt oracle/_ref)")
        return False
    out = os.path.join(DST, "mistral_inference")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    n = 0
    for name in sorted(os.listdir(pkg)):
        if name.endswith(".py"):
            py_compile.compile(os.path.join(pkg, name), cfile=os.path.join(out, name + "c"), doraise=True, optimize=0)
            n += 1
    with open(os.path.join(DST, "README"), "w") as f:
        f.write(f"sourceless bytecode of {pkg} ({n} modules), python {sys.version.split()[0]}; made by oracle/build_ref.py\n")
    if verbose:
        print(f"oracle/build_ref.py: {n} modules -> {out}")
    return True


if __name__ == "__main__":
    build()
