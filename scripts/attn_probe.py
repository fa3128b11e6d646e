"""Prefill attention alone at Mistral-7B dims: T tokens, 32 q heads / 8 kv heads, causal, first prefill (no ring)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip as h

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, Hkv, Dh, W = 32, 8, 128, 4096
qkv = torch.randn(T, (H + 2 * Hkv) * Dh, device="cuda").to(torch.bfloat16)
q_start = torch.tensor([0, T], dtype=torch.int32, device="cuda")
kv_before = torch.tensor([0], dtype=torch.int32, device="cuda")
f = lambda: h.attn_prefill(qkv, H, Hkv, Dh, None, None, W, q_start, kv_before, 1, T)
for _ in range(3): f()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): f()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / n * 1e3
pairs = sum(min(i + 1, W) for i in range(T))
print(f"attn_prefill T={T}: {us:.1f} us, {4 * H * Dh * pairs / us / 1e6:.1f} TF/s (causal-useful flops)")
