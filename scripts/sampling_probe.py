#!/usr/bin/env python3
"""Time the two sampling kernels alone on LM-head-like logits (fp32, scale of a random-init model): the greedy sample of the
launch path and the nucleus draw (temperature 0.7, top-p 0.8 / 0.95) at the vocabularies of Mistral-7B (32768) and Mistral-Nemo
(131072), batch 1 and 3 - microseconds per launch (median of 5 x 50 launches, HIP events).
    gpurun -- 'python scripts/sampling_probe.py'"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import torch  # noqa: E402
from mistral_inference import _hip  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(3)


def timed(f, reps=5, inner=50):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


for V in (32768, 131072):
    for B in (1, 3):
        for scale in (0.6, 3.0):  # random-init logits (|x| <~ 3) and a peaked, trained-model-like row
            x = (torch.randn(B, V, device=dev) * scale).float()
            g = timed(lambda: _hip.greedy_sample(x))
            line = f"V {V:6d} B {B} logit std {scale}: greedy {g:7.1f} us"
            for p in (0.8, 0.95):
                tok, lp = _hip.sample_top_p(x, 0.7, p, seed=1)
                t = timed(lambda: _hip.sample_top_p(x, 0.7, p, seed=1))
                line += f"   top-p {p}: {t:7.1f} us (token {int(tok[0])})"
            print(line, flush=True)
