#!/usr/bin/env python3
"""Static ISA audit of the hand-counted asm loads (cdna_hip_programming.md section 5.7 items 1 and 4).

An inline-asm load's destination registers are unprotected until OUR s_waitcnt: hipcc treats them as written at
;;#ASMEND and may copy, spill or reuse them while the data is still in flight.  This script compiles the given kernels
to gfx950 assembly and runs a forward dataflow over each kernel's control-flow graph with a FIFO model of the VM counter:

  * every VMEM instruction (asm or compiler issued, load or store) pushes an entry; asm loads carry their destination
    VGPRs, everything else carries none;
  * `s_waitcnt vmcnt(N)` (asm or compiler issued) keeps only the youngest N entries (loads retire in order; a store
    retiring out of order only makes the real state stricter than the model);
  * at control-flow joins the longer queue wins and registers are united position-wise from the young end (conservative);
  * any non-asm instruction that reads or writes a VGPR that is still in the queue is a FINDING.

It also requires scratch == 0 for kernels that use asm loads.  Exit code 1 on findings.

    python scripts/audit_asm_loads.py gemv engine attn_decode
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd", "csrc")
VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
        "flat_store", "scratch_load", "scratch_store")
QCAP = 48


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[:])v(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def parse_kernel(lines):
    """-> blocks: list of (label, [(is_asm, text)]), succ: dict label -> [labels]"""
    blocks, cur, label, in_asm = [], [], "__entry", False
    for raw in lines:
        t = raw.strip()
        if not t or t.startswith(("//",)) or (t.startswith(";") and not t.startswith(";;#ASM")):
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith("."):
            m = re.match(r"^(\.LBB[\w]+):", t)
            if m:
                blocks.append((label, cur))
                label, cur = m.group(1), []
            continue
        t = t.split(";")[0].strip()
        if t:
            cur.append((in_asm, t))
    blocks.append((label, cur))
    succ = {}
    for i, (lab, ins) in enumerate(blocks):
        nxt = blocks[i + 1][0] if i + 1 < len(blocks) else None
        out = []
        fall = True
        for _, t in ins:
            m = re.match(r"^s_(c?branch\w*)\s+(\.LBB\w+)", t)
            if m:
                out.append(m.group(2))
                if m.group(1) == "branch":
                    fall = False
        if ins and ins[-1][1].startswith("s_endpgm"):
            fall = False
        if fall and nxt:
            out.append(nxt)
        succ[lab] = out
    return blocks, succ


def join(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if len(a) < len(b):
        a, b = b, a
    res = [set(x) for x in a]
    for k in range(1, len(b) + 1):
        res[-k] |= b[-k]
    return tuple(frozenset(x) for x in res)


def transfer(state, ins, findings, lab):
    q = list(state)
    for is_asm, t in ins:
        op = t.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                q = q[len(q) - n:] if n < len(q) else q
            continue
        live = set().union(*q) if q else set()
        if not is_asm and live:
            hit = vregs(t) & live
            if hit:
                findings.add((lab, t, tuple(sorted(hit))[:4]))
        if op.startswith(VMEM):
            dst = frozenset(vregs(t.split(",")[0])) if (is_asm and "load" in op) else frozenset()
            q.append(dst)
            if len(q) > QCAP:
                q = q[-QCAP:]
    return tuple(q)


def audit_kernel(name, lines):
    blocks, succ = parse_kernel(lines)
    by = {lab: ins for lab, ins in blocks}
    uses_asm_loads = any(a and t.startswith("global_load") for _, ins in blocks for a, t in ins)
    if not uses_asm_loads:
        return None
    state_in = {blocks[0][0]: tuple()}
    work, findings = [blocks[0][0]], set()
    iters = 0
    while work and iters < 20000:
        iters += 1
        lab = work.pop()
        out = transfer(state_in[lab], by[lab], findings, lab)
        for s in succ.get(lab, []):
            if s not in by:
                continue
            new = join(state_in.get(s), out)
            if new != state_in.get(s):
                state_in[s] = new
                work.append(s)
    return findings


def main():
    bad = total = 0
    for src in sys.argv[1:]:
        sfile = f"/tmp/audit_{src}.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                        "-I" + CSRC, os.path.join(CSRC, src + ".hip"), "-o", sfile], check=True, capture_output=True)
        text = open(sfile).read()
        for km in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
            name, body = km.group(1), km.group(2).splitlines() + ["s_endpgm"]
            findings = audit_kernel(name, body)
            if findings is None:
                continue
            total += 1
            sc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b.*?\.amdhsa_private_segment_fixed_size (\d+)", text, re.S)
            scratch = int(sc.group(1)) if sc else 0
            if findings or scratch:
                bad += 1
                short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:80]
                print(f"[FINDING] {short}: scratch={scratch}, {len(findings)} instruction(s) touch in-flight asm-load registers")
                for f in sorted(findings)[:4]:
                    print("      ", f)
    print(f"audit: {total} kernels with asm loads, {'CLEAN' if not bad else str(bad) + ' with findings'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
