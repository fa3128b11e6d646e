"""The shipped headline build of the persistent decode engine (build_native.ENGINE_NEXT_FLAGS -> decode_engine_next.o) must not
carry compiler-inserted waits inside its loader's issue loops.

Background (profiles/EXPERIMENTS.md round 6): the loader wave issues its LDS-DMAs from inline asm and counts their completion
itself.  hipcc structurizes the kernel's role split (consumers / holders / loader) into a chain of `Flow` blocks, so its
wait-count pass carries the OTHER roles' outstanding register loads into the loader's code and guards the first write of every
such register with `s_waitcnt vmcnt(N)` - inside a loop that wait drains the DMA queue per layer / unit / fill: +2.7 % .. +30 %
per decode step, changing with every edit (the "regimes" of rounds 3-5).  scripts/engine_loader_waits.py finds such waits in the
ISA; this test holds the shipped flags to it ("check the `next` object's loader loop contains no
compiler-inserted vmcnt(0) or extra VMEM ops").  CPU only: hipcc cross-compiles."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))


def _report(extra=()):
    import tempfile
    import build_native as b
    import engine_loader_waits as w
    flags = [f for f in b.ENGINE_NEXT_FLAGS if not f.startswith("-DENG_SUFFIX")] + ["-DENG_SUFFIX=_chk", *extra]
    with tempfile.TemporaryDirectory() as d:
        reps, _ = w.analyse(w.compile_to_asm(flags, d))
    return reps


def test_shipped_headline_engine_has_no_compiler_waits_in_its_loader_loops():
    reps = _report()
    assert len(reps) == 1, [r["name"] for r in reps]      # ENG_HEADLINE_ONLY: one instantiation
    r = reps[0]
    assert r["dma"] >= 60                                  # the weight / K/V / LM-head streams are there
    in_loops = [x for x in r["suspicious"] if x[3] > 0]
    assert not in_loops, in_loops[:4]
    # every DMA is issued from inline asm (nothing for hipcc's pass to track): the builtin form names a 64-bit VGPR pair and
    # would show up here without the ASM markers
    assert r["builtin_dma"] <= 2, r["builtin_dma"]         # (the two asm statements of the per-lane-address K/V fallback)


def test_the_check_detects_a_known_bad_build():
    """The same source with the weight DMAs as hipcc builtins (ENG_SADDR=0) and without the stamp sites is the build rounds 3-5
    measured 14-19 % slow: the checker must see waits inside its loader loops."""
    reps = _report(extra=("-DENG_SADDR=0",))
    assert any(x[3] > 0 for r in reps for x in r["suspicious"])
