"""Kernel-boundary cost on this box: N dependent trivial launches, eager vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
dev = "cuda:0"
x = torch.randn(1, 256, device=dev).to(torch.bfloat16)
w = torch.ones(256, device=dev, dtype=torch.bfloat16)
out = torch.empty_like(x)
N = 2000
def run(n):
    for _ in range(n):
        _hip.rmsnorm(x, w, 1e-5, out=out)
run(100); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); run(N); e1.record(); t_host = time.perf_counter() - t0; e1.synchronize()
print(f"eager: {e0.elapsed_time(e1)*1e3/N:.2f} us/launch on GPU timeline, host enqueue {t_host*1e6/N:.2f} us/launch")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(10); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run(200)
    g.replay(); torch.cuda.synchronize()
    e0.record(s)
    for _ in range(10): g.replay()
    e1.record(s); e1.synchronize()
print(f"graph replay: {e0.elapsed_time(e1)*1e3/2000:.2f} us/launch")
