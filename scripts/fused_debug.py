import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-inference_amd"))
import torch
from mistral_inference import _hip
_hip.lib()
fn = C.CDLL(_hip.LIB_PATH).mi_debug_fused_ffn
vp = C.c_void_p
fn.argtypes = [vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]; fn.restype = C.c_int
dev = "cuda:0"; D, F = 4096, 14336
g = torch.Generator(device=dev).manual_seed(0)
def w(n, k): return ((torch.rand(n, k, generator=g, device=dev) * 2 - 1) * 0.02).to(torch.bfloat16)
W1, W3, W2 = w(F, D), w(F, D), w(D, F)
x0 = torch.randn(1, D, generator=g, device=dev).to(torch.bfloat16); nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
bar = torch.zeros(4096, dtype=torch.uint8, device=dev)
for nb in (8, 64, 512):
    ha, hb = x0.clone(), x0.clone()
    hid_a = torch.zeros(1, F, device=dev, dtype=torch.bfloat16); hid_b = torch.full((1, F), 7.0, device=dev, dtype=torch.bfloat16)
    _hip.linear(ha, (W1, W3), _hip.EPI_SWIGLU, norm_w=nw, eps=1e-5, out=hid_a)
    _hip.linear(hid_a, (W2,), _hip.EPI_RESIDUAL, residual=ha, out=ha)
    rc = fn(hb.data_ptr(), hb.data_ptr(), nw.data_ptr(), 1e-5, W1.data_ptr(), W3.data_ptr(), W2.data_ptr(), hid_b.data_ptr(), D, F,
            bar.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dh = (hid_a.float() - hid_b.float()).abs(); d = (ha.float() - hb.float()).abs()
    print(f"nb={nb}: rc={rc} hid: nan={int(torch.isnan(hid_b.float()).sum())} untouched(7.0)={int((hid_b == 7.0).sum())} maxdiff={float(dh[~torch.isnan(dh)].max())}"
          f" | h: nan={int(torch.isnan(hb.float()).sum())} maxdiff={float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1}")
    bad = torch.nonzero(torch.isnan(hb.float()).flatten())[:8].flatten().tolist()
    print("   first nan idx in h:", bad, " first bad hid idx:", torch.nonzero((dh > 0.01).flatten())[:8].flatten().tolist())
