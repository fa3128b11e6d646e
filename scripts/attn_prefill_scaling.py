#!/usr/bin/env python3
"""Where the prefill attention kernel's time goes, from outside: causal prompts of growing length (tile-units per CU grow
quadratically, blocks per CU linearly), the unmasked form (every block the same work, no diagonal tiles) and a sliding window
(every block the same work, window-edge + diagonal tiles) - microseconds per launch and per (256-query x 64-key) tile-unit.
    gpurun -- 'MI_ATTN_PREFILL_WAVES=8 python scripts/attn_prefill_scaling.py'"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import torch  # noqa: E402
from mistral_inference import _hip  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(7)
H, KV = 32, 8


def timed(f, reps=5, inner=10):
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


def units(T, W, causal):  # 256-query x 64-key tile-units the blocks of ONE head walk through (incl. their skipped / masked ones)
    n = 0
    for qt in range((T + 255) // 256):
        lo = max(0, qt * 256 - W + 1) if causal else 0
        hi = min(qt * 256 + 255, T - 1) if causal else T - 1
        n += (hi - lo + 64) // 64
    return n


for T, W, causal in [(2048, 2048, True), (4096, 4096, True), (6144, 6144, True), (8192, 8192, True), (12288, 12288, True), (16384, 16384, True),
                     (2048, 2048, False), (4096, 4096, False), (8192, 1024, True), (8192, 4096, True), (16384, 4096, True)]:
    qkv = (torch.randn(T, (H + 2 * KV) * 128, device=dev)).to(torch.bfloat16)
    q_start = torch.tensor([0, T], dtype=torch.int32, device=dev)
    kv_before = torch.tensor([0], dtype=torch.int32, device=dev)
    f = lambda: _hip.attn_prefill(qkv, H, KV, 128, None, None, W, q_start if causal else None, kv_before if causal else None, 1, T, causal=causal)  # noqa: E731
    us = timed(f)
    u = units(T, W, causal) * H / 256.0
    blocks = ((T + 255) // 256) * H / 256.0
    print(f"T {T:6d} W {W:6d} causal {int(causal)}: {us:9.1f} us  tile-units per CU {u:7.1f}  blocks per CU {blocks:4.1f}  us per tile-unit {us / u:6.3f}", flush=True)
