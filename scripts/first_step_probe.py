"""Where the K = 20 bracket loses ~20 us per step against the steady state: per-step HIP-event times of the first steps behind a
synchronisation, and the host's time per enqueued step.   python scripts/first_step_probe.py [idle_ms]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from mistral_inference.cache import BufferCache

dev = "cuda:0"
params = dict(bench.PRESETS["mistral-7b"][0])
model = bench.build_model(params, 0, 1, dev)
a = model.args
T0 = 4096
cache = BufferCache(model.n_local_layers, 1, T0 + 600, a.n_kv_heads, a.head_dim, a.sliding_window, device=dev, dtype=torch.bfloat16)
cache.reset()
prompt = torch.randint(0, a.vocab_size, (T0,), generator=torch.Generator().manual_seed(0)).to(dev)
with torch.inference_mode():
    logits = model.forward(prompt, [T0], cache)
    nxt = torch.argmax(logits[-1:], dim=-1)
    del logits
    sess = model.greedy_session(cache, nxt, graph=True)
    sess.run(8)
    sess.collect()
    stream = torch.cuda.current_stream(dev)
    for idle_ms in (0.0, 0.0, 1.0, 10.0):
        K = 12
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        sess.run(2)
        torch.cuda.synchronize()
        if idle_ms:
            time.sleep(idle_ms / 1e3)
        t0 = time.perf_counter()
        ev[0].record(stream)
        host = []
        for i in range(K):
            h0 = time.perf_counter()
            sess.run(1)
            host.append((time.perf_counter() - h0) * 1e6)
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        sess.collect()
        steps = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
        print(f"idle {idle_ms:5.1f} ms before the bracket: wall {wall / K * 1e3:.1f} us per step; per-step event us: "
              + " ".join(f"{x:.0f}" for x in steps) + " | host us per enqueue: " + " ".join(f"{x:.0f}" for x in host), flush=True)
