#!/usr/bin/env python3
"""Time the prefill attention kernel alone (causal, one sequence): Mistral-7B layer at 4096 tokens, Nemo at 8192, a 2048-token
prompt (128-query blocks) - microseconds (median of `reps` x 20 launches, HIP events), TFLOP/s, a digest of the output bits.
    gpurun -- 'MISTRAL_HIP_LIB=<variant .so> python scripts/attn_prefill_probe.py'"""
import hashlib
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import torch  # noqa: E402
from mistral_inference import _hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
torch.manual_seed(7)
for name, (T, H, KV) in {"7b_4096": (4096, 32, 8), "nemo_8192": (8192, 32, 8), "7b_2048": (2048, 32, 8), "7b_1000": (1000, 32, 8)}.items():
    qkv = (torch.randn(T, (H + 2 * KV) * 128, device=dev)).to(torch.bfloat16)
    q_start = torch.tensor([0, T], dtype=torch.int32, device=dev)
    kv_before = torch.tensor([0], dtype=torch.int32, device=dev)
    f = lambda: _hip.attn_prefill(qkv, H, KV, 128, None, None, T, q_start, kv_before, 1, T)  # noqa: E731
    out = f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    us = sorted(ts)[len(ts) // 2]
    flops = 4.0 * T * T * H * 128 / 2
    print(f"{name}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  sha {hashlib.sha1(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]}", flush=True)
