#!/usr/bin/env python3
"""Where the generic-storage path (csrc/generic.hip, mi_forward_generic) sits: Mistral-7B dims, a few layers, fp16 and fp32
storage - prefill tokens/s (TFLOP/s of the contraction + attention flops) and batch-1 decode ms/step (GB/s of the bytes a
step must read).  Not a BASELINE configuration (those are bf16 and run on the tuned kernels); this is the honest figure for
the compatibility path.

    gpurun --timeout 600 -- 'python scripts/generic_probe.py [layers] [prefill_tokens]'"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))


def main():
    import torch
    import bench
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer

    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    T0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    dev = "cuda:0"
    torch.cuda.set_device(0)
    out = {}
    for name, dtype in (("float16", torch.float16), ("float32", torch.float32)):
        params = dict(bench.MISTRAL_7B, n_layers=layers)
        args = TransformerArgs.from_dict(params)
        args.max_batch_size = 1
        with torch.device("meta"):
            model = Transformer(args)
        model = model.to(dtype).to_empty(device=dev)
        bench.init_weights_(model, seed=42)
        model._backend.invalidate()
        model.eval()
        es = 2 if dtype == torch.float16 else 4
        K = 16
        cache = BufferCache(layers, 1, T0 + K + 80, args.n_kv_heads, args.head_dim, args.sliding_window, device=dev, dtype=dtype)
        prompt = torch.randint(0, args.vocab_size, (T0,), generator=torch.Generator().manual_seed(0)).to(dev)
        with torch.inference_mode():
            cache.reset()
            model.forward(prompt, [T0], cache)
            ts = []
            for _ in range(3):
                cache.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                logits = model.forward(prompt, [T0], cache)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            pre = sorted(ts)[1]
            nxt = torch.argmax(logits[-1:], dim=-1)
            assert model._backend.generic
            sess = model.greedy_session(cache, nxt)
            sess.run(4)
            sess.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sess.run(K)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
            toks, _ = sess.collect()
        step_bytes = bench.decode_bytes_per_token(params, T0 + 4 + K // 2) * es // 2
        flops = bench.prefill_flops(params, T0)
        out[name] = {"layers": layers, "prefill_tokens": T0, "prefill_s": round(pre, 4), "prefill_tokens_per_s": round(T0 / pre, 1),
                     "prefill_tflops": round(flops / pre / 1e12, 1), "decode_ms_per_step": round(dt * 1e3, 4),
                     "decode_GBs": round(step_bytes / dt / 1e9, 1), "decode_frac_of_8TBs": round(step_bytes / dt / 8e12, 4),
                     "finite": bool(torch.isfinite(logits).all()), "tokens": toks[:4, 0].tolist()}
        print(name, json.dumps(out[name]), flush=True)
        del model, cache, sess, logits
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
