#!/usr/bin/env python3
"""Time the UNMODIFIED reference (mistral-inference, through oracle/shim) on the host cores: one batch-1 decode step at a
given context, on a bounded sample (a few of the model's layers + the LM head), scaled linearly in the layer count.

Test/bench infrastructure (bench.py's `cpu_baseline` leg runs it in a subprocess when the reference source is present -
the build container - or, on the GPU box, its byte-compiled form under oracle/_ref, see oracle/build_ref.py).  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")
if not os.path.isdir(os.path.join(REF, "mistral_inference")):
    REF = os.path.join(HERE, "_ref")  # the same package byte-compiled from the unmodified source (oracle/build_ref.py)
sys.path[:0] = [os.path.join(HERE, "shim"), REF]

import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", required=True, help="JSON of the model params (full model)")
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--sample-layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    opt = ap.parse_args()
    from mistral_inference.args import TransformerArgs   # the reference
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    import mistral_inference
    assert os.path.realpath(mistral_inference.__file__).startswith(os.path.realpath(REF)), "not the reference package"

    full = json.loads(opt.params)
    p = dict(full, n_layers=opt.sample_layers)
    args = TransformerArgs.from_dict(p)
    args.max_batch_size = 1
    torch.manual_seed(0)
    model = Transformer(args).to(torch.bfloat16).eval()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    W = p.get("sliding_window") or opt.ctx

    def fresh_cache():
        c = BufferCache(args.n_layers, 1, opt.ctx + opt.steps + 4, args.n_kv_heads, args.head_dim, p.get("sliding_window"))
        c.to(device="cpu", dtype=torch.bfloat16)
        c.reset()
        return c

    tok = torch.tensor([1])
    with torch.inference_mode():
        cache = fresh_cache()
        prompt = torch.randint(0, args.vocab_size, (opt.ctx,), generator=torch.Generator().manual_seed(0))
        t0 = time.perf_counter()
        model.forward(prompt, seqlens=[opt.ctx], cache=cache)   # fills the rings exactly as generate() would
        t_prefill = time.perf_counter() - t0
        best = (float("inf"), 1)
        for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            model.forward(tok, seqlens=[1], cache=cache)
            t0 = time.perf_counter()
            model.forward(tok, seqlens=[1], cache=cache)
            best = min(best, (time.perf_counter() - t0, n))
        cores = best[1]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        for _ in range(opt.steps):
            model.forward(tok, seqlens=[1], cache=cache)
        t_full = (time.perf_counter() - t0) / opt.steps
        h = torch.randn(1, args.dim).to(torch.bfloat16)
        t0 = time.perf_counter()
        for _ in range(opt.steps):
            model.output(model.norm(h)).float()
        t_head = (time.perf_counter() - t0) / opt.steps
    per_layer = max(1e-9, (t_full - t_head) / opt.sample_layers)
    t_model = t_head + full["n_layers"] * per_layer
    print(json.dumps({
        "value": round(1.0 / t_model, 3), "unit": "tokens/s", "cores": cores, "kind": "reference",
        "sample": f"unmodified reference (oracle/shim for xformers/simple_parsing) decode step at ctx {opt.ctx} (W={W}) with "
                  f"{opt.sample_layers} of {full['n_layers']} layers + LM head, {opt.steps} steps, bf16, {cores} threads; per-layer "
                  f"time x{full['n_layers']} + head ({per_layer * 1e3:.1f} ms/layer, {t_head * 1e3:.1f} ms head; "
                  f"{opt.sample_layers}-layer prefill of {opt.ctx} tokens took {t_prefill:.1f} s)"}))


if __name__ == "__main__":
    main()
