"""Does the weight-streaming GEMV run faster when its weights are resident in the 256 MiB Infinity Cache?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
dev = "cuda:0"
D, F, L = 4096, 14336, 12
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k): return ((torch.rand(n, k, generator=g, device=dev) * 2 - 1) * 0.02).to(torch.bfloat16)
W1 = [w(F, D) for _ in range(L)]; W3 = [w(F, D) for _ in range(L)]; W2 = [w(D, F) for _ in range(L)]; WO = [w(D, D) for _ in range(L)]
x = torch.randn(1, D, device=dev).to(torch.bfloat16); nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
hid = torch.randn(1, F, device=dev).to(torch.bfloat16); res = torch.zeros(1, D, device=dev, dtype=torch.bfloat16)
out_f = torch.empty(1, F, device=dev, dtype=torch.bfloat16); out_d = torch.empty(1, D, device=dev, dtype=torch.bfloat16)
def timeit(fn, n):
    fn(0); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            for i in range(n): fn(i)
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): graph.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
def w13(i, cyc): 
    j = i % L if cyc else 0
    _hip.linear(x, (W1[j], W3[j]), _hip.EPI_SWIGLU, norm_w=nw, eps=1e-5, out=out_f)
def w2(i, cyc):
    j = i % L if cyc else 0
    _hip.linear(hid, (W2[j],), _hip.EPI_RESIDUAL, residual=res, out=out_d)
def wo(i, cyc):
    j = i % L if cyc else 0
    _hip.linear(x, (WO[j],), _hip.EPI_RESIDUAL, residual=res, out=out_d)
for name, fn, mb in (("W1|W3 235MB", w13, 234.9), ("W2 117MB", w2, 117.4), ("Wo 33.5MB", wo, 33.6)):
    a = timeit(lambda i: fn(i, True), 48); b = timeit(lambda i: fn(i, False), 48)
    print(f"{name}: cycling {L} layers {a:7.2f} us ({mb/a*1e3/1e3:.2f} TB/s) | same layer (cache-resident) {b:7.2f} us ({mb/b*1e3/1e3:.2f} TB/s)")
# two layers alternating for W2 (2 x 117 MB = 235 MB, fits MALL)
c = timeit(lambda i: _hip.linear(hid, (W2[i % 2],), _hip.EPI_RESIDUAL, residual=res, out=out_d), 48)
print(f"W2 alternating 2 layers (235 MB working set): {c:7.2f} us")
