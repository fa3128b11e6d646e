"""Feasibility: does prefetching the NEXT op's weights into the Infinity Cache on a side stream shorten a chain of
weight-streaming GEMVs?  Chain per 'layer': W1|W3 GEMV -> W2 GEMV (different weights every layer)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
dev = "cuda:0"
D, F, L = 4096, 14336, 16
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k): return ((torch.rand(n, k, generator=g, device=dev) * 2 - 1) * 0.02).to(torch.bfloat16)
W1 = [w(F, D) for _ in range(L)]; W3 = [w(F, D) for _ in range(L)]; W2 = [w(D, F) for _ in range(L)]
x = torch.randn(1, D, device=dev).to(torch.bfloat16); nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
hid = torch.empty(1, F, device=dev, dtype=torch.bfloat16); h = torch.zeros(1, D, device=dev, dtype=torch.bfloat16)
sink = torch.zeros(8, device=dev)
def c13(j): _hip.linear(x, (W1[j], W3[j]), _hip.EPI_SWIGLU, norm_w=nw, eps=1e-5, out=hid)
def c2(j): _hip.linear(hid, (W2[j],), _hip.EPI_RESIDUAL, residual=h, out=h)
def prefetch(t): sink[0:1].add_(t.view(torch.int32).view(-1)[::16384 // 4].sum())  # placeholder, replaced below
# a real streaming prefetch: read every byte with a cheap reduction kernel (allocates in MALL)
def prefetch(t): torch.sum(t.view(torch.int32), dtype=torch.int64)
def build(overlap):
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(sA):
        with torch.cuda.graph(gph, stream=sA):
            for j in range(L):
                if overlap:
                    ev = torch.cuda.Event(); ev.record(sA); sB.wait_event(ev)
                    with torch.cuda.stream(sB): prefetch(W2[j])          # overlaps the W1|W3 GEMV
                c13(j)
                if overlap:
                    ev = torch.cuda.Event(); ev.record(sA); sB.wait_event(ev)
                    with torch.cuda.stream(sB):
                        prefetch(W1[(j + 1) % L]); prefetch(W3[(j + 1) % L])   # overlaps the W2 GEMV
                c2(j)
            if overlap:
                ev = torch.cuda.Event(); ev.record(sB); sA.wait_event(ev)
    return gph, sA
for overlap in (False, True, False, True):
    gph, s = build(overlap)
    with torch.cuda.stream(s):
        gph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): gph.replay()
        e1.record(s); e1.synchronize()
    print(f"overlap={overlap}: {e0.elapsed_time(e1) * 1e3 / (5 * L):.2f} us per (W1|W3 + W2) pair")
