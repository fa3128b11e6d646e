#!/usr/bin/env python3
"""Fastest possible check of an engine build: full-size model, engine vs launch path bit for bit (6 steps), then ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mistral_inference import _hip
from mistral_inference.cache import BufferCache

t00 = time.time()
m = bench.build_model(dict(bench.MISTRAL_7B), 0, 1, "cuda")
a = m.args
T, steps = 4096, 6
ids = torch.randint(0, a.vocab_size, (T + steps,), generator=torch.Generator().manual_seed(0)).cuda()

def run(engine):
    prev = _hip.set_decode_engine(engine)
    c = BufferCache(m.n_local_layers, 1, T + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=torch.bfloat16)
    c.reset()
    with torch.inference_mode():
        m.forward(ids[:T], [T], c)
        outs = [m.forward(ids[T + i:T + i + 1], [1], c)[0].clone() for i in range(steps)]
    torch.cuda.synchronize()
    _hip.set_decode_engine(prev)
    return outs, c

ref, c0 = run(False)
got, c1 = run(True)
st = _hip.decode_engine_status(m._backend._workspace)
eq = all(torch.equal(x, y) for x, y in zip(ref, got))
n = min(c0.cache_sizes[0], T + steps)
eqr = all(torch.equal(c0.cache_k[l][:, :n], c1.cache_k[l][:, :n]) and torch.equal(c0.cache_v[l][:, :n], c1.cache_v[l][:, :n]) for l in range(m.n_local_layers))
print("status", st, "logits bit-equal", eq, "rings bit-equal", eqr, flush=True)
# timing: 48 eager engine steps (same token)
nxt = ids[-1:].clone()
with torch.inference_mode():
    for _ in range(4):
        m.forward(nxt, [1], c1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(48):
        m.forward(nxt, [1], c1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 48
print(f"engine eager ms/step {dt * 1e3:.4f}  total script {time.time() - t00:.1f}s", flush=True)
