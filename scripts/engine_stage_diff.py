#!/usr/bin/env python3
"""Where do the persistent decode engine and the launch path first differ?  (debug aid)

Both paths run the same decode steps from identical cache states; after each step the LAST layer's intermediate
vectors are read back - from the workspace buffers of the launch path (csrc/api.hip `carve`) and from the engine's
hand-off granules - and compared bit for bit: q|k|v, attention output, hidden (SwiGLU), residual stream.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import bench  # noqa: E402
from mistral_inference import _hip  # noqa: E402
from mistral_inference.cache import BufferCache  # noqa: E402


def al(x, a=256):
    return (x + a - 1) // a * a


def layout(p):
    D, H, Hkv, F = p["dim"], p["n_heads"], p["n_kv_heads"], p["hidden_dim"]
    R = H // Hkv
    g = {}
    off = 0
    for name, n in (("h", D // 2), ("qkv", (H + 2 * Hkv) * 64), ("att", H * 64), ("h1", D // 2), ("hid", F // 2),
                    ("part", Hkv * 32 * R * 130)):
        g[name] = (off, n)
        off += n
    gran_bytes = off * 8
    ws = {}
    o = 4096
    ws["gran"] = o
    o += al(gran_bytes)
    for name, nbytes in (("xn", D * 2), ("qkv", (H + 2 * Hkv) * 128 * 2), ("attn", H * 128 * 2), ("hid", F * 2)):
        ws[name] = (o, nbytes)
        o += al(nbytes)
    return g, ws


def gran_bf16(wsb, base, off, n):
    raw = wsb[base + off * 8: base + (off + n) * 8].view(torch.int32).view(-1, 2)  # [n, (value, tag)]
    return raw[:, 0].contiguous().view(torch.bfloat16)  # 2 bf16 per granule


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, nargs="+", default=[8, 6, 4, 2, 1])
    ap.add_argument("--steps", type=int, default=5)
    opt = ap.parse_args()
    T = 4096
    for nl in opt.layers:
        params = dict(bench.MISTRAL_7B)
        params["n_layers"] = nl
        model = bench.build_model(params, 0, 1, "cuda")
        a = model.args
        g, ws = layout(params)
        ids = torch.randint(0, a.vocab_size, (T + opt.steps,), generator=torch.Generator().manual_seed(0)).cuda()
        caches = []
        for _ in range(2):
            c = BufferCache(nl, 1, T + opt.steps + 2, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=torch.bfloat16)
            c.reset()
            caches.append(c)
        with torch.inference_mode():
            _hip.set_decode_engine(False)
            model.forward(ids[:T], [T], caches[0])
            for l in range(nl):
                caches[1].cache_k[l].copy_(caches[0].cache_k[l])
                caches[1].cache_v[l].copy_(caches[0].cache_v[l])
            caches[1].kv_seqlens = caches[0].kv_seqlens.clone()
            caches[1]._seen = list(caches[0]._seen)
            for s in range(opt.steps):
                tok = ids[T + s:T + s + 1]
                _hip.set_decode_engine(False)
                lo_a = model.forward(tok, [1], caches[0]).clone()
                torch.cuda.synchronize()
                wsb = model._backend._workspace
                ref = {k: wsb[ws[k][0]: ws[k][0] + ws[k][1]].clone().view(torch.bfloat16) for k in ("qkv", "attn", "hid")}
                _hip.set_decode_engine(True)
                lo_b = model.forward(tok, [1], caches[1]).clone()
                torch.cuda.synchronize()
                wsb = model._backend._workspace
                got = {"qkv": gran_bf16(wsb, ws["gran"], *g["qkv"]), "attn": gran_bf16(wsb, ws["gran"], *g["att"]),
                       "hid": gran_bf16(wsb, ws["gran"], *g["hid"]), "h1": gran_bf16(wsb, ws["gran"], *g["h1"]),
                       "h": gran_bf16(wsb, ws["gran"], *g["h"])}
                line = [f"layers={nl} step={s}"]
                for k in ("qkv", "attn", "hid"):
                    d = (ref[k] != got[k]).nonzero().flatten().tolist()
                    line.append(f"{k}: {len(d)} diff" + (f" first idx {d[0]} ref {float(ref[k][d[0]]):.6g} got {float(got[k][d[0]]):.6g}" if d else ""))
                dl = (lo_a != lo_b).sum().item()
                line.append(f"logits: {dl} diff")
                rk = sum(int((caches[0].cache_k[l] != caches[1].cache_k[l]).sum()) for l in range(nl))
                rv = sum(int((caches[0].cache_v[l] != caches[1].cache_v[l]).sum()) for l in range(nl))
                line.append(f"rings: k {rk} v {rv} diff")
                print(" | ".join(line), flush=True)
        del model, caches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
