#!/usr/bin/env python3
"""Timeline of one decode step on the persistent decode engine (debug / tuning aid).

Registers a trace buffer (mi_debug_set_engine_trace), runs one batch-1 decode step of the BASELINE configs[1] model at
context 4096 and prints, per phase of a layer, when consumer wave 0 of every CU passed it (mean / min / max over CUs,
averaged over the middle layers) and how long the loader took per weight segment.  Raw stamps: gpurun_out/engine_trace.npy
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import bench  # noqa: E402
from mistral_inference import _hip  # noqa: E402
from mistral_inference.cache import BufferCache  # noqa: E402

CONS = ["start", "h gathered", "normed", "qkv rows", "cbar", "q gathered", "att pieces", "partial pub", "merge gathered",
        "merge pub", "attn gathered", "wo rows", "h1 gathered", "normed2", "w13 rows", "hid gathered", "w2 rows", "end"]
LOAD = ["L start", "L qkv", "L kv", "L wo", "L w13", "L w2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--prefill", type=int, default=4096)
    ap.add_argument("--model", default="mistral-7b", choices=sorted(bench.PRESETS))
    opt = ap.parse_args()
    params = dict(bench.PRESETS[opt.model][0])
    params["n_layers"] = min(opt.layers, 32)   # (the trace buffer holds one launch = 32 layers)
    opt.layers = params["n_layers"]
    model = bench.build_model(params, 0, 1, "cuda")
    a = model.args
    cache = BufferCache(model.n_local_layers, 1, opt.prefill + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda",
                        dtype=torch.bfloat16)
    cache.reset()
    ids = torch.randint(0, a.vocab_size, (opt.prefill,), generator=torch.Generator().manual_seed(0)).cuda()
    L = _hip.lib()
    # the stamp sites live in the frozen default build, the wide and the MoE build; the headline shape's `next` build is compiled
    # without them (ENG_TRACE=0) - route dense models to the frozen build for the trace (variant 2)
    if not params.get("moe"):
        L.mi_debug_set_engine_variant(2)
    with torch.inference_mode():
        nxt = torch.argmax(model.forward(ids, [opt.prefill], cache)[-1:], dim=-1)
        for _ in range(4):
            nxt = torch.argmax(model.forward(nxt, [1], cache), dim=-1)
        torch.cuda.synchronize()
        nbytes = L.mi_debug_engine_trace_bytes()
        buf = torch.zeros(nbytes // 8, dtype=torch.int64, device="cuda")
        L.mi_debug_set_engine_trace(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nxt = torch.argmax(model.forward(nxt, [1], cache), dim=-1)
        e1.record()
        torch.cuda.synchronize()
        L.mi_debug_set_engine_trace(None)
    print("status", _hip.decode_engine_status(model._backend._workspace), "step ms (traced)", e0.elapsed_time(e1))
    t = buf.cpu().numpy().reshape(-1, 32, 26).astype(np.float64)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "engine_trace.npy"), t)
    report(t, opt.layers)


def report(t, nl):
    """Print the phase timeline of one traced launch: t[cu][layer][event] in 100 MHz ticks (scripts/engine_ab.py calls this too)."""
    nb = t.shape[0]
    us = 0.01  # 100 MHz ticks -> us
    t0 = t[:, 0, 0].min()
    print(f"CUs {nb}, layers {nl}; whole launch span (first start -> last end): "
          f"{(t[:, nl - 1, 17].max() - t0) * us:.1f} us; per layer {(t[:, nl - 1, 17].max() - t0) * us / nl:.1f} us")
    mid = range(2, max(3, nl - 2))
    print("\nconsumer wave 0: time of each event relative to the layer's earliest start (us), averaged over middle layers")
    print(f"{'event':16s} {'mean':>8s} {'min':>8s} {'max':>8s}   {'delta(mean)':>10s}")
    prev = None
    for ev, name in enumerate(CONS):
        rel = np.stack([(t[:, l, ev] - t[:, l, 0].min()) * us for l in mid])  # [layers, cus]
        valid = np.stack([t[:, l, ev] > 0 for l in mid])
        m = rel[valid].mean() if valid.any() else float("nan")
        lo = np.where(valid, rel, np.inf).min(axis=1).mean()
        hi = np.where(valid, rel, -np.inf).max(axis=1).mean()
        d = m - prev if prev is not None else 0.0
        print(f"{name:16s} {m:8.2f} {lo:8.2f} {hi:8.2f}   {d:10.2f}")
        prev = m
    print("\nloader: time of each event relative to the same origin (us)")
    for ev, name in enumerate(LOAD):
        rel = np.stack([(t[:, l, 18 + ev] - t[:, l, 0].min()) * us for l in mid])
        print(f"{name:16s} {rel.mean():8.2f} {rel.min(axis=1).mean():8.2f} {rel.max(axis=1).mean():8.2f}")
    stalls = t[:, :nl, 24]
    print("loader ring-full stalls per CU (cumulative at last layer): mean %.1f max %.0f" % (stalls[:, nl - 1].mean(), stalls[:, nl - 1].max()))
    if nl > 6:
        d13 = ((t[:, 2:nl - 2, 14] - t[:, 2:nl - 2, 13]) * us).mean(axis=1)   # W1|W3 phase of consumer wave 0, per CU
        d2 = ((t[:, 2:nl - 2, 16] - t[:, 2:nl - 2, 15]) * us).mean(axis=1)
        print("W1|W3 phase by XCD (cu % 8), us:", " ".join(f"{d13[x::8].mean():.2f}" for x in range(8)),
              "| W2:", " ".join(f"{d2[x::8].mean():.2f}" for x in range(8)))
    seg = (t[:, 2:nl - 2, 23] - t[:, 2:nl - 2, 18]) * us
    print("loader time per layer: mean %.2f us, min %.2f, max %.2f  (1.70 MB per CU per layer -> %.1f GB/s per CU, %.2f TB/s chip)"
          % (seg.mean(), seg.min(), seg.max(), 1.70e6 / seg.mean() / 1e3, 1.70e6 / seg.mean() / 1e3 * nb / 1e3))
    handoffs(t, nl)


def handoffs(t, nl):
    """Every hand-off of a layer split into the part that is SKEW (how long after the median CU the slowest producer
    finished its phase) and the part that is LATENCY (first consumer through its sweep - last producer done), from the
    stamps of consumer wave 0 of every CU.  Usable offline: `engine_trace.py --analyze gpurun_out/engine_trace.npy`."""
    us = 0.01
    mid = range(2, max(3, nl - 2))
    edges = [("W2 rows (prev) -> h", 17, 1, True),  # (event 17: all four waves are through their W2 units; 16 is wave 0 alone)
             ("q|k|v rows -> q", 3, 5, False), ("partials -> merge", 7, 8, False),
             ("merged -> attn", 9, 10, False), ("Wo rows -> h1", 11, 12, False), ("W1|W3 rows -> hid", 14, 15, False)]
    print("\nhand-offs (us, middle layers): producers' skew max-median | first sweep done - last producer done | "
          "median consumer's wait (median gathered - median producer end)")
    total = 0.0
    for name, pe, ce, prev in edges:
        sk, lat, wait = [], [], []
        for l in mid:
            p = t[:, l - 1, pe] if prev else t[:, l, pe]
            c = t[:, l, ce]
            ok = (p > 0) & (c > 0)
            if not ok.any():
                continue
            p, c = p[ok] * us, c[ok] * us
            sk.append(p.max() - np.median(p))
            lat.append(c.min() - p.max())
            wait.append(np.median(c) - np.median(p))
        if wait:
            print(f"  {name:20s} {np.mean(sk):6.2f} | {np.mean(lat):6.2f} | {np.mean(wait):6.2f}")
            total += np.mean(wait)
    print(f"  sum of the median waits: {total:.1f} us per layer")


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--analyze":
        _t = np.load(sys.argv[2])
        handoffs(_t, _t.shape[1])
    else:
        main()
