"""[W1|W3 GEMV -> W2 GEMV] as two launches vs one persistent launch with a grid barrier (weights prefetched across it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
lib = _hip.lib()
fn = C.CDLL(_hip.LIB_PATH).mi_debug_fused_ffn
vp = C.c_void_p
fn.argtypes = [vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
fn.restype = C.c_int
dev = "cuda:0"
D, F, L = 4096, 14336, 12
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k): return ((torch.rand(n, k, generator=g, device=dev) * 2 - 1) * 0.02).to(torch.bfloat16)
W1 = [w(F, D) for _ in range(L)]; W3 = [w(F, D) for _ in range(L)]; W2 = [w(D, F) for _ in range(L)]
x0 = torch.randn(1, D, generator=g, device=dev).to(torch.bfloat16); nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
hid = torch.empty(1, F, device=dev, dtype=torch.bfloat16)
bar = torch.zeros(4096, dtype=torch.uint8, device=dev)
def two(h, j):
    _hip.linear(h, (W1[j], W3[j]), _hip.EPI_SWIGLU, norm_w=nw, eps=1e-5, out=hid)
    _hip.linear(hid, (W2[j],), _hip.EPI_RESIDUAL, residual=h, out=h)
def fused(h, j, nb):
    rc = fn(h.data_ptr(), h.data_ptr(), nw.data_ptr(), 1e-5, W1[j].data_ptr(), W3[j].data_ptr(), W2[j].data_ptr(),
            hid.data_ptr(), D, F, bar.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
# correctness: same chain both ways
ha, hb = x0.clone(), x0.clone()
for j in range(L): two(ha, j)
for j in range(L): fused(hb, j, 512)
torch.cuda.synchronize()
print("abort flag:", int(bar.view(torch.int32)[(8*16 + 16 + 8*16)]), " max |two - fused| =", (ha.float() - hb.float()).abs().max().item(), " |h| max", ha.float().abs().max().item())
def timeit(fn_, n=L * 4):
    h = x0.clone(); fn_(h, 0); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gph, stream=s):
            for i in range(n): fn_(h, i % L)
        gph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): gph.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
print(f"two launches : {timeit(two):7.2f} us per FFN")
for nb in (256, 512, 768, 1024):
    print(f"fused nb={nb:4d}: {timeit(lambda h, j: fused(h, j, nb)):7.2f} us per FFN")
