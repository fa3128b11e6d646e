"""Race screen: the MFMA kernels keep LDS-DMA / register prefetch in flight across barriers; a hazard there shows up as a
run-to-run difference.  Every kernel is launched N times on the same inputs and compared bit for bit on the device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip as h

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g, device="cuda") * scale).to(BF)
bad = 0
for name, M, K, Nn, epi in (("qkv", 4096, 4096, 6144, h.EPI_STORE), ("wo", 4096, 4096, 4096, h.EPI_RESIDUAL),
                            ("w13", 4096, 4096, 14336, h.EPI_SWIGLU), ("w2", 4096, 14336, 4096, h.EPI_RESIDUAL),
                            ("lm", 2048, 4096, 32768, h.EPI_LOGITS), ("ragged", 1000, 1024, 5000 // 8 * 8, h.EPI_STORE)):
    x, w = rnd(M, K), rnd(Nn, K, scale=K ** -0.5)
    ws = (w, rnd(Nn, K, scale=K ** -0.5)) if epi == h.EPI_SWIGLU else (w,)
    res = rnd(M, Nn) if epi == h.EPI_RESIDUAL else None
    ref = h.linear(x, ws, epi, residual=res)
    diff = sum(int(not torch.equal(h.linear(x, ws, epi, residual=res), ref)) for _ in range(N))
    print(f"gemm {name}: {diff} of {N} launches differ", flush=True)
    bad += diff
for T, H, Hkv in ((4096, 32, 8), (1500, 32, 8), (700, 8, 2)):
    qkv = rnd(T, (H + 2 * Hkv) * 128)
    qs = torch.tensor([0, T], dtype=torch.int32, device="cuda"); kb = torch.zeros(1, dtype=torch.int32, device="cuda")
    f = lambda: h.attn_prefill(qkv, H, Hkv, 128, None, None, 4096, qs, kb, 1, T)
    ref = f()
    diff = sum(int(not torch.equal(f(), ref)) for _ in range(N))
    print(f"attn_prefill T={T}: {diff} of {N} launches differ", flush=True)
    bad += diff
x, w = rnd(4096, 4096), rnd(32768, 4096, scale=1 / 64)
tgt = torch.randint(0, 32768, (4096,), device="cuda", dtype=torch.int32)
ref = h.lm_head_logprobs(x, w, tgt)
diff = sum(int(not torch.equal(h.lm_head_logprobs(x, w, tgt), ref)) for _ in range(N // 4))
print(f"lm_head_logprobs: {diff} of {N // 4} launches differ")
bad += diff
print("SOAK", "CLEAN" if bad == 0 else f"{bad} DIFFERENCES")
sys.exit(1 if bad else 0)
