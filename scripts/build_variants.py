#!/usr/bin/env python3
"""Build A/B variants of libmistral_hip.so for scripts/gpu_ab.sh (compile-time switches of the decode engine).

    python scripts/build_variants.py engine          # every engine variant below (or name the ones wanted)
    gpurun --timeout 900 -- 'bash scripts/gpu_ab.sh 2'
    python scripts/build_variants.py gemm            # the prefill-GEMM variants (FILE_VARIANTS; only gemm256.o differs)
    gpurun --timeout 900 -- 'python scripts/prefill_probe.py'

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

# Variants of ONE source file: only that object is rebuilt, the rest is linked from the main build.
# (gemm256.hip switches: G256_PRIO, G256_BAL, G256_DMA_AFTER, G256_ABL - see the top of that file)
FILE_VARIANTS = {
    "g_prio1": ("gemm256.hip", ("-DG256_PRIO=1",)),
    "g_prio2": ("gemm256.hip", ("-DG256_PRIO=2",)),
    "g_dma_after": ("gemm256.hip", ("-DG256_DMA_AFTER=1",)),
    "g_bal": ("gemm256.hip", ("-DG256_BAL=1",)),
    "g_bal_prio1": ("gemm256.hip", ("-DG256_BAL=1", "-DG256_PRIO=1")),
    "g_bal_prio2": ("gemm256.hip", ("-DG256_BAL=1", "-DG256_PRIO=2")),
    "g_bal_after": ("gemm256.hip", ("-DG256_BAL=1", "-DG256_DMA_AFTER=1")),
    "g_bal_after_prio1": ("gemm256.hip", ("-DG256_BAL=1", "-DG256_DMA_AFTER=1", "-DG256_PRIO=1")),
    "g_bal2": ("gemm256.hip", ("-DG256_BAL=2",)),            # + the DMAs only in the two light read segments
    "g_bal2_prio1": ("gemm256.hip", ("-DG256_BAL=2", "-DG256_PRIO=1")),
    "g_bal2_after": ("gemm256.hip", ("-DG256_BAL=2", "-DG256_DMA_AFTER=1")),
    "g_bal2_after_prio1": ("gemm256.hip", ("-DG256_BAL=2", "-DG256_DMA_AFTER=1", "-DG256_PRIO=1")),
    "g_pipe1": ("gemm256.hip", ("-DG256_PIPE=1",)),          # software-pipelined loop, one barrier per K tile
    "g_pipe2": ("gemm256.hip", ("-DG256_PIPE=2",)),
    "g_pipe1_prio2": ("gemm256.hip", ("-DG256_PIPE=1", "-DG256_PRIO=2")),
    # engine: more tickets in the compiler-flag lottery, each ON TOP of the shipped max-memory-clause strategy
    "e_nocluster": ("decode_engine.hip", ("-mllvm", "-misched-cluster=0")),
    "e_trackers": ("decode_engine.hip", ("-mllvm", "-amdgpu-use-amdgpu-trackers=1")),
    "e_prera_topdown": ("decode_engine.hip", ("-mllvm", "-misched-prera-direction=topdown")),
    "e_prera_bottomup": ("decode_engine.hip", ("-mllvm", "-misched-prera-direction=bottomup")),
    "e_postra_bottomup": ("decode_engine.hip", ("-mllvm", "-misched-postra-direction=bottomup")),
    "e_nopostmisched": ("decode_engine.hip", ("-mllvm", "-enable-post-misched=0")),
    "e_noslp": ("decode_engine.hip", ("-fno-slp-vectorize",)),
    "e_nounroll": ("decode_engine.hip", ("-fno-unroll-loops",)),
    "e_relaxed_occ": ("decode_engine.hip", ("-mllvm", "-amdgpu-schedule-relaxed-occupancy=1")),
    # the stamp sites again, as a one-object variant (round 4: PMC counters of both builds, scripts/engine_pmc.sh)
    "e_trace0": ("decode_engine.hip", ("-DENG_TRACE=0",)),
    "e_trace2": ("decode_engine.hip", ("-DENG_TRACE=2",)),
    # which stamp sites matter (consumer events 0-17, loader events 18-25; bit set = site compiled in)
    "e_mask_loader": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x03fc0000u",)),
    "e_mask_cons": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x0003ffffu",)),
    "e_mask_cons_lo": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x000001ffu",)),
    "e_mask_cons_hi": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x0003fe00u",)),
    "e_mask_even": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x01555555u",)),
    "e_mask_odd": ("decode_engine.hip", ("-DENG_TRACE_MASK=0x02aaaaaau",)),
    # launch path (Nemo dims, batch > 1): the same strategy for the GEMV / decode-attention sources
    "l_gemv_mmc": ("gemv.hip", ("-mllvm", "-amdgpu-sched-strategy=max-memory-clause")),
    "l_attn_mmc": ("attn_decode.hip", ("-mllvm", "-amdgpu-sched-strategy=max-memory-clause")),
    "l_gemv_ilp": ("gemv.hip", ("-mllvm", "-amdgpu-sched-strategy=max-ilp")),
    "a_vb0": ("attn_prefill.hip", ("-DATT_VB128=0",)),          # V^T fragments as two 8-byte reads (rounds 1-6)
    "a_fast0": ("attn_prefill.hip", ("-DATT_FASTLOAD=0",)),    # every tile through the general staging path (rounds 1-5)
    "a_trace2": ("attn_prefill.hip", ("-DATT_TRACE=2",)),      # + cycles per phase of waves 0 and 7
    "a_trace": ("attn_prefill.hip", ("-DATT_TRACE=1",)),       # per-block timeline: scripts/attn_prefill_trace.py
    "a_prio1": ("attn_prefill.hip", ("-DATT_PRIO=1",)),
    "a_prio2": ("attn_prefill.hip", ("-DATT_PRIO=2",)),
    "g_abl_nodma": ("gemm256.hip", ("-DG256_ABL=1",)),      # timing ablations: WRONG results by construction
    "g_abl_noreads": ("gemm256.hip", ("-DG256_ABL=2",)),
    "g_abl_nomfma": ("gemm256.hip", ("-DG256_ABL=3",)),
    "g_abl_mfmaonly": ("gemm256.hip", ("-DG256_ABL=4",)),
    "g_abl_mfmaonly32": ("gemm256.hip", ("-DG256_ABL=5",)),
    "g_clk": ("gemm256.hip", ("-DG256_CLK=1",)),              # + shader clock measured across block 0's main loop
    "g_clk_abl_mfmaonly": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_ABL=4")),
    "g_clk_abl_nomfma": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_ABL=3")),
    "g_clk_abl_dmaonly": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_ABL=7")),
    "g_clk_abl_readsonly": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_ABL=8")),
    "g_clk_dma_after": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_DMA_AFTER=1")),
    "g_clk_pipe2": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_PIPE=2")),
    "g_clk_split": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_SPLIT=1")),
    "g_clk_bal_after": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_BAL=1", "-DG256_DMA_AFTER=1")),
    "g_clk_bal2_after": ("gemm256.hip", ("-DG256_CLK=1", "-DG256_BAL=2", "-DG256_DMA_AFTER=1")),
    "g_abl_mfmaonly32x8": ("gemm256.hip", ("-DG256_ABL=6",)),
}


# Engine experiment SLOTS: many compiles of decode_engine.hip (headline instantiation only, ~8 s each) linked into ONE library
# under the names *_x<N>; scripts/engine_ab.py times them all in one process on one set of weights (mi_debug_set_engine_slot).
#     python scripts/build_variants.py engine_slots [names...]   ->  lib/variants/libmistral_hip_slots.so + slots.json
_AP = ("-DENG_ABORT_RARE=1", "-DENG_CONS_PRIO=1")   # round-5 calls 1-2: abort word read rarely (-0.8 %), consumers at priority 1 (-0.25 %)
_NS = _AP + ("-DENG_HOLD_STAGE=2", "-DENG_SADDR=2")
_CE = ("-DENG_CLEAN_ENTRY=1",)                     # round 6: the loader's wait-count scoreboard emptied at its entry (decode_engine.hip)
_N0 = _NS + ("-DENG_TRACE=0",)                      # = build_native.ENGINE_NEXT_FLAGS (the shipped headline build) without its suffix
ENGINE_SLOTS = {
    "copy": (),
    "nx": _AP + ("-DENG_HOLD_STAGE=2",),
    "ns": _NS,
    "ns_trace0": _NS + ("-DENG_TRACE=0",),                          # = round-5 call 9's decode_engine_next.o
    "nst_hid": _NS + ("-DENG_TRACE=0", "-DENG_NOSTOP=32"),          # loader not stopped during the hid sweep (ring empty there)
    "nst_hid_attn": _NS + ("-DENG_TRACE=0", "-DENG_NOSTOP=40"),
    "nst_hid_h": _NS + ("-DENG_TRACE=0", "-DENG_NOSTOP=33"),
    "ns_hid": _NS + ("-DENG_NOSTOP=32",),
    # a sweep of the remaining knobs on top of the shipped flags with the head-major rings (the K/V issue is 2.2 us now, not 5)
    "hid_q": _N0 + ("-DENG_NOSTOP=34",),
    "hid_attn": _N0 + ("-DENG_NOSTOP=40",),
    "hid_h1": _N0 + ("-DENG_NOSTOP=48",),
    "hid_merge": _N0 + ("-DENG_NOSTOP=36",),
    "hid_stage1": _AP + ("-DENG_HOLD_STAGE=1", "-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32"),
    "hid_stage3": _AP + ("-DENG_HOLD_STAGE=3", "-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32"),
    "hid_prio2": ("-DENG_ABORT_RARE=1", "-DENG_CONS_PRIO=2", "-DENG_HOLD_STAGE=2", "-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32"),
    "ns_ce_hid": _NS + _CE + ("-DENG_NOSTOP=32",),                  # the shipped flags WITH stamp sites, clean by the static check: timelines
    "nst_hid_stage3": _AP + ("-DENG_SADDR=2", "-DENG_TRACE=0", "-DENG_NOSTOP=32"),
    "nst_hid_hold4": _NS + ("-DENG_TRACE=0", "-DENG_NOSTOP=32", "-DENG_SLP_HOLD=4"),
    # round 6 (the K/V-phase experiments of this round - asm K/V pieces, fine / tight publication, a reordered stream - are kept as
    # scripts/probes/decode_engine_round6_kv_experiments.patch; what shipped is the head-major ring layout)
    "ce": _N0 + _CE,                                                # + a modelled wait at the loader's entry
    "ce_hid": _N0 + _CE + ("-DENG_NOSTOP=32",),
    "ns_ce": _NS + _CE,                                             # (with the stamp sites: timelines)
    "ns_hid": _NS + ("-DENG_NOSTOP=32",),
}


def build_engine_slots(names):
    import json
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    names = [n for n in (names or ENGINE_SLOTS)]
    obj_dir = "/tmp/obj_slots"
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = b._hipcc()
    src = os.path.join(b.CSRC, "decode_engine.hip")
    jobs, objs = [], []
    for i, n in enumerate(names):
        flags = ENGINE_SLOTS[n] if n in ENGINE_SLOTS else tuple(n.split(","))  # (an ad-hoc slot: comma-separated flags)
        o = os.path.join(obj_dir, f"de_x{i}.o")
        objs.append(o)
        jobs.append([hipcc, *b.FLAGS, *b.PER_FILE_FLAGS.get("decode_engine.hip", []), f"-DENG_SUFFIX=_x{i}", "-DENG_HEADLINE_ONLY=1",
                     *flags, "-c", src, "-o", o])
    api_o = os.path.join(obj_dir, "api.o")
    slot_list = " ".join(f"X({i})" for i in range(len(names)))
    jobs.append([hipcc, *b.FLAGS, f"-DMI_SLOT_LIST={slot_list}", "-c", os.path.join(b.CSRC, "api.hip"), "-o", api_o])
    with ThreadPoolExecutor(max_workers=8) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-3000:])
    main = [api_o if s_ == "api.hip" else os.path.join(b.OBJ, s_.replace(".hip", ".o")) for s_ in b.SOURCES]
    main += [os.path.join(b.OBJ, v) for v in b.VARIANT_OBJECTS]
    lib = os.path.join(ROOT, "lib", "variants", "libmistral_hip_slots.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *main, *objs, "-ldl", "-o", lib], check=True)
    with open(os.path.join(ROOT, "lib", "variants", "slots.json"), "w") as f:
        json.dump({"slots": [{"index": i, "name": n, "flags": list(ENGINE_SLOTS.get(n, n.split(",")))} for i, n in enumerate(names)]}, f, indent=1)
    return lib


def build_file_variant(name, src, flags):
    import subprocess
    obj_dir = f"/tmp/obj_{name}"
    os.makedirs(obj_dir, exist_ok=True)
    o = os.path.join(obj_dir, src.replace(".hip", ".o"))
    hipcc = b._hipcc()
    subprocess.run([hipcc, *b.FLAGS, *b.PER_FILE_FLAGS.get(src, []), *flags, "-c", os.path.join(b.CSRC, src), "-o", o], check=True)
    objs = [o if s == src else os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES]
    objs += [os.path.join(b.OBJ, v) for v in b.VARIANT_OBJECTS]  # (the wide engine object: always the main build's)
    lib = os.path.join(ROOT, "lib", "variants", f"libmistral_hip_{name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", lib], check=True)
    return lib


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "lib", "variants"), exist_ok=True)
    b.build(verbose=False)
    if "engine_slots" in sys.argv[1:]:
        print(build_engine_slots(sys.argv[sys.argv.index("engine_slots") + 1:]), flush=True)
        sys.exit(0)
    for name, (src, flags) in FILE_VARIANTS.items():
        if name in sys.argv[1:] or "gemm" in sys.argv[1:] and name.startswith("g_") or "attn" in sys.argv[1:] and name.startswith("a_") or "engine_flags" in sys.argv[1:] and name.startswith("e_") and not name.startswith("e_mask") or "engine_masks" in sys.argv[1:] and name.startswith("e_mask") or "launch_flags" in sys.argv[1:] and name.startswith("l_"):
            print(name, build_file_variant(name, src, flags), flush=True)
    for name, flags in VARIANTS.items():
        if name not in sys.argv[1:] and "engine" not in sys.argv[1:]:
            continue
        print(name, b.build(verbose=False, extra_flags=flags, obj_dir=f"/tmp/obj_{name}",
                            lib=os.path.join(ROOT, "lib", "variants", f"libmistral_hip_{name}.so")), flush=True)
