#!/usr/bin/env python3
"""Same-process A/B of decode-engine builds: ONE model, ONE box, every engine variant timed in turn, interleaved, with a
bit-equality check of each variant against the frozen shipped object.

    python scripts/build_variants.py engine_slots [names...]     # here: lib/variants/libmistral_hip_slots.so + slots.json
    gpurun --timeout 600 -- 'MISTRAL_HIP_LIB=$PWD/mistral-inference_amd/lib/variants/libmistral_hip_slots.so \
                             python scripts/engine_ab.py --steps 200 --reps 3 --trace 2'

Why not scripts/gpu_ab.sh: a bench.py process per library spends ~25 s on weights and prefill for 0.2 s of decode, and boxes
differ by more than most changes.  Here a variant costs a graph capture + `steps` x 2.7 ms, so dozens fit one call, and every
number is from the same weights, rings and clocks.  Entries: `frozen` (MI_ENGINE_VARIANT=2 routing: the shipped default
object), `routed` (the library's own choice: decode_engine_next.o for the headline shape) and every slot.
Output: gpurun_out/engine_ab.log (+ engine_ab.json, + timelines of the best `--trace` entries and of `frozen`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import bench  # noqa: E402
import engine_trace  # noqa: E402
from mistral_inference import _hip  # noqa: E402
from mistral_inference.cache import BufferCache  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--prefill", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--check-steps", type=int, default=12, help="greedy steps of the bit-equality check")
    ap.add_argument("--trace", type=int, default=0, help="print the in-kernel timeline of the N fastest entries (and of `frozen`)")
    ap.add_argument("--only", default=None, help="comma-separated entry names")
    ap.add_argument("--trace-names", default=None, help="print the timeline of these entries (comma-separated) as well")
    ap.add_argument("--knobs", default=None, help="extra entries `<slot>@thin1|thin0|depth3`, comma-separated")
    opt = ap.parse_args()
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, "engine_ab.log"), "w")

    def say(*a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()

    L = _hip.lib()
    slots = []
    meta = os.path.join(os.path.dirname(_hip.LIB_PATH), "slots.json")
    if hasattr(L, "mi_debug_set_engine_slot") and os.path.exists(meta):
        slots = json.load(open(meta))["slots"]
    entries = [("frozen", None), ("routed", None)] + [(s["name"], s["index"]) for s in slots]
    # run-time loader knobs on top of a slot: "<slot>@thin1" (one fill in flight during sweeps), "<slot>@depth3"
    by_name = dict(entries)
    for extra in (opt.knobs.split(",") if opt.knobs else []):
        base = extra.split("@")[0]
        if base in by_name:
            entries.append((extra, by_name[base]))
    if opt.only:
        keep = set(opt.only.split(","))
        entries = [e for e in entries if e[0] in keep or e[0] == "frozen"]

    def select(name, idx):
        if slots:
            L.mi_debug_set_engine_slot(-1 if idx is None else idx)
        L.mi_debug_set_engine_variant(2 if name == "frozen" else 0)
        knob = name.split("@")[1] if "@" in name else ""
        L.mi_debug_set_engine_knobs(1 if knob == "thin1" else (0 if knob == "thin0" else 2), 3 if knob == "depth3" else 2)

    params = dict(bench.PRESETS["mistral-7b"][0])
    if opt.layers:
        params["n_layers"] = opt.layers
    model = bench.build_model(params, 0, 1, "cuda")
    a = model.args
    T0 = opt.prefill
    cache = BufferCache(model.n_local_layers, 1, T0 + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=torch.bfloat16)
    cache.reset()
    ids = torch.randint(0, a.vocab_size, (T0,), generator=torch.Generator().manual_seed(0)).cuda()
    step_bytes = bench.decode_bytes_per_token(params, T0 + 64)
    with torch.inference_mode():
        first = torch.argmax(model.forward(ids, [T0], cache)[-1:], dim=-1)
        torch.cuda.synchronize()
        snap_k = {i: cache.cache_k[i].clone() for i in range(cache.n_layers)}
        snap_v = {i: cache.cache_v[i].clone() for i in range(cache.n_layers)}
        snap_len, snap_seen = cache.kv_seqlens.clone(), list(cache._seen)

        def restore():
            for i in range(cache.n_layers):
                cache.cache_k[i].copy_(snap_k[i])
                cache.cache_v[i].copy_(snap_v[i])
            cache.kv_seqlens.copy_(snap_len)
            cache._seen = list(snap_seen)
            torch.cuda.synchronize()

        # ---- bit-equality: the same `check_steps` greedy steps from the same state on every entry
        ref = None
        verdict = {}
        for name, idx in entries:
            select(name, idx)
            restore()
            sess = model.greedy_session(cache, first, graph=False)
            sess.run(opt.check_steps)
            toks, lps = sess.collect()
            got = (toks.clone(), lps.clone(), sess.logits.clone())
            st = _hip.decode_engine_status(model._backend._workspace)
            assert st["status"] == 0, (name, st)
            if ref is None:
                ref = got
                verdict[name] = "reference"
            else:
                verdict[name] = "bit-equal" if all(torch.equal(x, y) for x, y in zip(got, ref)) else "DIFFERENT"
            del sess
        say("bit-equality vs `frozen` over", opt.check_steps, "greedy steps (tokens, log-probs, last logits row):",
            ", ".join(f"{n}: {v}" for n, v in verdict.items()))

        # ---- timing: reps interleaved passes
        restore()
        times = {n: [] for n, _ in entries}
        for rep in range(opt.reps):
            order = entries if rep % 2 == 0 else entries[::-1]
            for name, idx in order:
                select(name, idx)
                sess = model.greedy_session(cache, first, graph=True)
                sess.run(opt.warmup)
                sess.collect()
                sess.run(2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sess.run(opt.steps)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                sess.collect()
                times[name].append(dt / opt.steps * 1e3)
                del sess
        base = float(np.median(times["frozen"]))
        rows = sorted(((float(np.median(v)), n) for n, v in times.items()))
        say(f"\n{'entry':24s} {'median ms':>10s} {'vs frozen':>10s} {'% of 8 TB/s':>12s}   runs")
        for med, n in rows:
            say(f"{n:24s} {med:10.4f} {100 * (med / base - 1):+9.2f}% {100 * step_bytes / (med * 1e-3) / 8e12:11.2f}%   "
                + " ".join(f"{x:.4f}" for x in times[n]) + ("" if verdict.get(n) in ("bit-equal", "reference") else "   <-- " + str(verdict.get(n))))
        json.dump({"times_ms": times, "bit_equal": verdict, "slots": slots, "steps": opt.steps, "bytes_per_step": step_bytes},
                  open(os.path.join(out_dir, "engine_ab.json"), "w"), indent=1)

        # ---- timelines
        if (opt.trace > 0 or opt.trace_names) and params["n_layers"] <= 32:
            want = ["frozen"] + [n for _, n in rows if n != "frozen"][: opt.trace]
            want += [n for n in (opt.trace_names.split(",") if opt.trace_names else []) if n in dict(entries) and n not in want]
            nbytes = L.mi_debug_engine_trace_bytes()
            for name in want:
                idx = dict(entries)[name]
                select(name, idx)
                sess = model.greedy_session(cache, first, graph=False)
                sess.run(6)
                sess.collect()
                buf = torch.zeros(nbytes // 8, dtype=torch.int64, device="cuda")
                L.mi_debug_set_engine_trace(buf.data_ptr())
                sess.run(1)
                torch.cuda.synchronize()
                L.mi_debug_set_engine_trace(None)
                sess.collect()
                t = buf.cpu().numpy().reshape(-1, 32, 26).astype(np.float64)
                np.save(os.path.join(out_dir, f"engine_trace_{name}.npy"), t)
                say(f"\n================ timeline: {name}")
                import contextlib
                import io
                cap = io.StringIO()
                with contextlib.redirect_stdout(cap):
                    engine_trace.report(t, params["n_layers"])
                say(cap.getvalue())
                del sess
    select("routed", None)


if __name__ == "__main__":
    main()
