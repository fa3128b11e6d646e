"""Runs a fixed, seeded set of weight-streaming / attention / whole-stack calls through whichever libmistral_hip the
environment selects (MISTRAL_HIP_LIB) and saves every output.  Used by test_gpu_safe_variant.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mistral-inference_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from mistral_inference import _hip  # noqa: E402

BF = torch.bfloat16
out = {}


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).cuda()


reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for rep in range(reps):
    for (K, N) in ((4096, 1536), (14336, 1024), (1024, 5120), (256, 130)):
        for M in (1, 2, 3, 4, 6, 8):
            x = rnd(M, K, seed=100 * rep + M)
            w1, w3 = rnd(N, K, seed=7 + rep, scale=K ** -0.5), rnd(N, K, seed=8 + rep, scale=K ** -0.5)
            res = rnd(M, N, seed=9 + rep)
            nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(BF).cuda()
            out[f"store.{rep}.{K}.{N}.{M}"] = _hip.linear(x, (w1,), _hip.EPI_STORE)
            out[f"resid.{rep}.{K}.{N}.{M}"] = _hip.linear(x, (w1,), _hip.EPI_RESIDUAL, residual=res)
            out[f"swiglu.{rep}.{K}.{N}.{M}"] = _hip.linear(x, (w1, w3), _hip.EPI_SWIGLU, norm_w=nw, eps=1e-5)
            out[f"logits.{rep}.{K}.{N}.{M}"] = _hip.linear(x, (w1,), _hip.EPI_LOGITS, norm_w=nw, eps=1e-5)
    for (H, Hkv) in ((4, 4), (4, 2), (32, 8), (48, 8), (16, 2)):
        for W, lens in ((300, [1, 299, 300]), (4096, [4096, 17, 5000]), (1100, [1100])):
            B = len(lens)
            g = torch.Generator().manual_seed(rep * 17 + W + H)
            ck = torch.randn(B, W, Hkv, 128, generator=g).to(BF).cuda()
            cv = torch.randn(B, W, Hkv, 128, generator=g).to(BF).cuda()
            q = torch.randn(B, H * 128, generator=g).to(BF).cuda()
            pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32).cuda()
            out[f"attn.{rep}.{H}.{Hkv}.{W}"] = _hip.attn_decode(q, ck, cv, H, pos)

# whole stack: dense and MoE tiny models, prefill + decode steps (exercises QKV_ROPE, MoE GEMVs, ring writes)
import mistral_oracle as mo  # noqa: E402
from hip_util import write_checkpoint  # noqa: E402
from mistral_inference.generate import generate  # noqa: E402
from mistral_inference.transformer import Transformer  # noqa: E402
import tempfile  # noqa: E402

for name, extra in (("dense", {}), ("moe", dict(num_experts=8, num_experts_per_tok=2))):
    args = mo.OracleArgs(dim=512, n_layers=2, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
                         vocab_size=1000, sliding_window=16, **extra)
    with tempfile.TemporaryDirectory() as d:
        model = Transformer.from_folder(write_checkpoint(d, args, mo.synth_weights(args, seed=5)), max_batch_size=3,
                                        device="cuda", dtype=BF)
    toks, lps = generate([[1, 2, 3, 4, 5, 6, 7, 8, 9], [4, 5], [9, 8, 7, 6]], model, max_tokens=24, temperature=0.0)
    out[f"gen.{name}.tokens"] = torch.tensor(toks)
    out[f"gen.{name}.logprobs"] = torch.tensor([lp[-24:] for lp in lps], dtype=torch.float64)
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
print("saved", len(out), "tensors; lib =", _hip.LIB_PATH)
