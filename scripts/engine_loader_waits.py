#!/usr/bin/env python3
"""Static check of a decode-engine build: does hipcc's wait-count pass drain the LOADER's DMA queue?

    python scripts/engine_loader_waits.py                 # the shipped headline build (build_native.ENGINE_NEXT_FLAGS)
    python scripts/engine_loader_waits.py -DENG_KVX=5 ... # the shipped flags + these
    python scripts/engine_loader_waits.py --flags "<all flags>"   # exactly these (an experiment slot)

The loader wave (csrc/decode_engine.hip: run_loader) issues its LDS-DMAs from inline asm and counts their completion itself
(`s_waitcnt vmcnt(0 | 16 | 32 | 47)` statements of fill_begin / fill_end / flush).  hipcc does not see those loads - but it sees its
own, and it structurizes the kernel's role split into a chain of `Flow` blocks, so the loader's code is statically reachable
from the holder / consumer code: wherever the loader first WRITES a VGPR that a load of another role (or an earlier load of
its own) may still name, the pass inserts `s_waitcnt vmcnt(N)`.  Inside the issue loops such a wait drains the DMA queue on every
pass: +2.7 % (a drain per layer) to +30 % (a drain per fill) per decode step, and which registers collide changes with every
edit - the "regimes" of rounds 3-5 (profiles/EXPERIMENTS.md, round 6).  This script compiles the build with -save-temps, finds
the loader's region in the ISA (between the first and the last LDS-DMA) and lists every wait there that is neither one of the
loader's own statements nor the wait behind the abort poll (a load whose value is used at once), with the loop depth of its
basic block - per kernel of the object.  Exit status 1 if one sits inside a loop (waits executed once per launch are harmless).
Builds whose loader uses the builtin DMA (the default, wide and MoE objects) legitimately carry hipcc's own counted waits for
those loads; for them the listing is a diagnostic, not a verdict.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import build_native as b  # noqa: E402


def compile_to_asm(flags, workdir):
    src = os.path.join(b.CSRC, "decode_engine.hip")
    cmd = [b._hipcc(), *b.FLAGS, *b.PER_FILE_FLAGS.get("decode_engine.hip", []), *flags, "-c", src, "-o", os.path.join(workdir, "de.o"),
           "--save-temps=obj"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return os.path.join(workdir, f)
    raise RuntimeError("no device assembly produced")


def loop_depth(L, i):
    """Loop depth of the basic block that holds line i (hipcc annotates block labels with `in Loop: Header=... Depth=N`)."""
    j = i
    while j > 0 and not re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):?", L[j].strip()):
        j -= 1
    for k in range(j, min(j + 4, len(L))):
        m = re.search(r"Depth=(\d+)", L[k])
        if m:
            return int(m.group(1))
    return 0


def analyse_kernel(L, name):
    """One kernel's lines -> dict(name, dma, builtin_dma, suspicious=[(line, text, context, loop depth)])"""
    dma = [i for i, l in enumerate(L) if "global_load_lds_dwordx4" in l]
    if not dma:
        return None
    lo, hi = dma[0] - 400, dma[-1] + 200  # (the loader's code follows the consumers' and the holders' in every kernel)
    wait = re.compile(r"s_waitcnt\s+vmcnt\((\d+)\)")
    bad = []
    for i in range(max(lo, 1), min(hi, len(L) - 1)):
        m = wait.search(L[i])
        if not m:
            continue
        if "ASMSTART" in L[i - 1]:  # one of the loader's own statements
            continue
        nxt = L[i + 1].strip()
        prev_load = any("global_load_dword " in L[j] and "sc1" in L[j] for j in range(max(i - 6, 0), i))
        if m.group(1) == "0" and nxt.startswith("v_cmp_eq_u32") and prev_load:  # the abort poll (every 1024th spin)
            continue
        bad.append((i + 1, L[i].strip(), " | ".join(x.strip() for x in L[i - 2:i + 3]), loop_depth(L, i)))
    builtin = [i + 1 for i in dma if re.search(r"global_load_lds_dwordx4 v\[\d+:\d+\]", L[i])]
    return {"name": name, "dma": len(dma), "builtin_dma": len(builtin), "suspicious": bad}


def analyse(path):
    """-> (per-kernel reports, resource line)"""
    L = open(path).read().split("\n")
    starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w*decode_engine_kernel\w*:", l)]
    if not starts:
        raise RuntimeError("no decode_engine_kernel in the object")
    reps = []
    for k, st in enumerate(starts):
        en = starts[k + 1] if k + 1 < len(starts) else len(L)
        for j in range(st, en):
            if L[j].startswith(".Lfunc_end"):
                en = j
                break
        r = analyse_kernel(L[st:en], L[st].split(":")[0])
        if r:
            reps.append(r)
    res = [l.strip() for l in L if re.search(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):", l)]
    return reps, " ".join(res[-4:])


def main(argv):
    if "--flags" in argv:
        flags = argv[argv.index("--flags") + 1].split()
    else:
        flags = [f for f in b.ENGINE_NEXT_FLAGS if not f.startswith("-DENG_SUFFIX")] + ["-DENG_SUFFIX=_chk"] + [a for a in argv if a.startswith("-")]
    with tempfile.TemporaryDirectory() as d:
        reps, resources = analyse(compile_to_asm(flags, d))
    print(f"flags: {' '.join(flags)}")
    print(f"last kernel's resources: {resources}")
    total_in_loops = 0
    for rep in reps:
        in_loops = [x for x in rep["suspicious"] if x[3] > 0]
        total_in_loops += len(in_loops)
        print(f"{rep['name']}: {rep['dma']} LDS-DMA instructions ({rep['builtin_dma']} in the VGPR-address form); compiler-inserted "
              f"waits in the loader's region: {len(rep['suspicious'])}, of them inside a loop: {len(in_loops)}")
        for ln, text, ctx, depth in rep["suspicious"][: (40 if "-v" in argv else 6)]:
            print(f"    +{ln} (loop depth {depth}): {text}    [{ctx}]")
    print(f"waits inside loops of a loader's region (a drain of the DMA queue per layer / unit / fill): {total_in_loops}")
    return 1 if total_in_loops else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
