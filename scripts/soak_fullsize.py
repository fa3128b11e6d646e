"""Full-size soak (BASELINE configs[1]: Mistral-7B dims, 32 layers, sliding window 4096): prefill 4096 tokens, then N greedy decode
steps - across the wrap of the 4096-slot rings - once on the persistent engine and once on the launch path: every token and the
last logits row must be IDENTICAL, the log-probabilities equal up to the fp32 summation order of the log-sum-exp (the engine
reduces it inside the LM head's sweep, the launch path in greedy_rows_kernel: tests/test_gpu_greedy.py holds them to 2e-5)
(reference loop: generate.py:120-140; both paths restate it).
   python scripts/soak_fullsize.py [steps] [preset] [prefill] [layers] [engine variant]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from mistral_inference import _hip
from mistral_inference.cache import BufferCache

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
name = sys.argv[2] if len(sys.argv) > 2 else "mistral-7b"     # bench.PRESETS key
T0 = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # (mixtral-8x22b: 7 = one of the 8 pipeline stages)
variant = int(sys.argv[5]) if len(sys.argv) > 5 else -1        # mi_debug_set_engine_variant (3: the opt-in Nemo build)
dev = "cuda:0"
params = dict(bench.PRESETS[name][0])
if layers:
    params["n_layers"] = layers
model = bench.build_model(params, 0, 1, dev)
a = model.args
if variant >= 0:
    _hip.lib().mi_debug_set_engine_variant(variant)
print(f"{bench.PRESETS[name][1]} dims, {params['n_layers']} layers, prefill {T0}, {N} steps, engine variant {variant}", flush=True)
prompt = torch.randint(0, a.vocab_size, (T0,), generator=torch.Generator().manual_seed(1)).to(dev)
out = {}
with torch.inference_mode():
    for engine in (True, False):
        prev = _hip.set_decode_engine(engine)
        cache = BufferCache(model.n_local_layers, 1, T0 + N + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device=dev, dtype=torch.bfloat16)
        cache.reset()
        logits = model.forward(prompt, [T0], cache)
        nxt = torch.argmax(logits[-1:], dim=-1)
        del logits
        sess = model.greedy_session(cache, nxt)
        toks, lps = [], []
        t0 = time.perf_counter()
        left = N
        while left > 0:
            n = min(left, 500)
            sess.run(n)
            t, l = sess.collect()
            toks.append(t.cpu())
            lps.append(l.cpu())
            left -= n
        dt = time.perf_counter() - t0
        st = _hip.decode_engine_status(model._backend._workspace)
        out[engine] = (torch.cat(toks), torch.cat(lps), sess.logits.clone().cpu())
        print(f"engine={engine}: {N} steps in {dt:.2f} s ({N / dt:.1f} tokens/s incl. collects); status {st}", flush=True)
        _hip.set_decode_engine(prev)
te, le, ge = out[True]
tl, ll, gl = out[False]
same_t, same_g = torch.equal(te, tl), torch.equal(ge, gl)
lp_err = float((le - ll).abs().max())
same_l = lp_err < 2e-5
first = int((te != tl).any(dim=1).nonzero()[0, 0]) if not same_t else None
print(f"positions {T0} .. {T0 + N} (sliding_window {a.sliding_window}); distinct tokens {te.unique().numel()}; "
      f"tokens identical: {same_t} (first difference at step {first}); log-probabilities max |difference| {lp_err:.2e} ({int((le != ll).sum())} of {le.numel()} differ in the last bits); last logits row identical: {same_g}")
sys.exit(0 if (same_t and same_l and same_g) else 1)
