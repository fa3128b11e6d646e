"""Times the prefill GEMM shapes of Mistral-7B through mi_linear (hip events, 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip as h

def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
BF = torch.bfloat16
shapes = [("qkv", 4096, 6144, h.EPI_STORE), ("wo", 4096, 4096, h.EPI_RESIDUAL), ("w13", 4096, 14336, h.EPI_SWIGLU),
          ("w2", 14336, 4096, h.EPI_RESIDUAL), ("lm", 4096, 32768, h.EPI_LOGITS)]
for name, K, N, epi in shapes:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(BF)
    ws = (w, w.clone()) if epi == h.EPI_SWIGLU else (w,)
    res = torch.zeros(M, N, device="cuda", dtype=BF) if epi == h.EPI_RESIDUAL else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == h.EPI_LOGITS else BF)
    us = t(lambda: h.linear(x, ws, epi, residual=res, out=out))
    fl = 2.0 * M * K * N * (2 if epi == h.EPI_SWIGLU else 1)
    print(f"{name:4s} M={M} K={K} N={N}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
if len(sys.argv) > 2:  # fixed-cost fit: N = 4096, K sweep, per epilogue
    for epi, nm in ((h.EPI_STORE, "store"), (h.EPI_RESIDUAL, "resid"), (h.EPI_LOGITS, "logit")):
        for K in (512, 2048, 4096, 8192):
            N = 4096
            x = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
            w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(BF)
            res = torch.zeros(M, N, device="cuda", dtype=BF) if epi == h.EPI_RESIDUAL else None
            out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == h.EPI_LOGITS else BF)
            us = t(lambda: h.linear(x, (w,), epi, residual=res, out=out))
            print(f"{nm} K={K}: {us:8.1f} us", flush=True)
if len(sys.argv) > 3:  # per-round cost of the SWIGLU form: N = 2048 k gives k rounds of 256 tiles
    for N in (2048, 4096, 8192, 14336):
        K = 4096
        x = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
        w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(BF)
        out = torch.empty(M, N, device="cuda", dtype=BF)
        us = t(lambda: h.linear(x, (w, w), h.EPI_SWIGLU, out=out))
        print(f"swiglu N={N}: {us:8.1f} us", flush=True)
