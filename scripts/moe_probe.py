import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
dev = "cuda:0"
D, F = 4096, 14336
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k, s): return ((torch.rand(n, k, generator=g, device=dev) * 2 - 1) * s).to(torch.bfloat16)
gate = w(8, D, 1/64)
x = torch.randn(4096, D, generator=g, device=dev).to(torch.bfloat16)
nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
xn = _hip.rmsnorm(x, nw, 1e-5)
idx, ww = _hip.moe_router(xn, gate, 2)
print("expert token counts:", torch.bincount(idx.flatten().long(), minlength=8).tolist())
W1, W3 = w(F, D, 1/64), w(F, D, 1/64)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (128, 512, 1024, 2048, 4096):
    xm = xn[:M].contiguous()
    us = t(lambda: _hip.linear(xm, (W1, W3), _hip.EPI_SWIGLU))
    print(f"dense swiglu GEMM M={M}: {us:8.1f} us  {M*F*D*4/us/1e6:7.1f} TF")
