"""Parity AT DEPTH: the full 32-layer BASELINE configs[1] model (Mistral-7B-v0.3 dims, random init) against the CPU oracle
in bf16 AND against an fp32 run of the oracle on the same (bf16-representable) weights.

SURVEY.md section 7 asks for three numbers, because a max-abs tolerance between two bf16 implementations means nothing
without knowing how far bf16 itself sits from the fp32 answer:

    e_hip = max |HIP - oracle_fp32|      e_o16 = max |oracle_bf16 - oracle_fp32|      d = max |HIP - oracle_bf16|

The assertion is relative: the HIP path may not be further from the fp32 truth than the reference's own bf16 arithmetic
is (x 1.25 on the maximum, x 1.1 on the mean), at 32 layers, on a 256-token prompt plus 8 teacher-forced decode steps
(the batch-1 decode steps run on the persistent decode engine), once with the BASELINE window (4096) and once with
sliding_window = 128 so that the ring wraps inside the prompt and keeps wrapping during decode.

The oracle runs LAYER-MAJOR (a 32-stage pipeline of one-layer OracleModels, the reference's own pipeline contract,
transformer.py:94-98): weights are generated per layer on the host, copied into the HIP model, used by the four oracle
passes (2 windows x 2 dtypes) and dropped - 7.2 G parameters never sit in host memory at once.
"""
import math
import time

import pytest
import torch
import torch.nn.functional as F

import mistral_oracle as mo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

P7B = dict(dim=4096, n_layers=32, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
           vocab_size=32768, rope_theta=1e6)
PROMPT, STEPS = 256, 8


def _lin(o, i, g):
    return ((torch.rand(o, i, generator=g) * 2 - 1) / math.sqrt(i)).to(BF)


def _layer_weights(l, p, g):
    D, Fh, nq, nkv = p["dim"], p["hidden_dim"], p["n_heads"] * p["head_dim"], p["n_kv_heads"] * p["head_dim"]
    pre = f"layers.{l}."
    return {
        pre + "attention.wq.weight": _lin(nq, D, g), pre + "attention.wk.weight": _lin(nkv, D, g),
        pre + "attention.wv.weight": _lin(nkv, D, g), pre + "attention.wo.weight": _lin(D, nq, g),
        pre + "attention_norm.weight": (1 + 0.1 * torch.randn(D, generator=g)).to(BF),
        pre + "ffn_norm.weight": (1 + 0.1 * torch.randn(D, generator=g)).to(BF),
        pre + "feed_forward.w1.weight": _lin(Fh, D, g), pre + "feed_forward.w2.weight": _lin(D, Fh, g),
        pre + "feed_forward.w3.weight": _lin(Fh, D, g),
    }


def test_32_layers_vs_bf16_and_fp32_oracle():
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(P7B)
    L, V, D = p["n_layers"], p["vocab_size"], p["dim"]
    oargs = mo.OracleArgs.from_params(p)
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda").eval()
    sd = dict(model.named_parameters())
    g = torch.Generator().manual_seed(1234)
    emb = torch.randn(V, D, generator=g).to(BF)
    final_norm = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
    out_w = _lin(V, D, g)
    with torch.no_grad():
        sd["tok_embeddings.weight"].copy_(emb)
        sd["norm.weight"].copy_(final_norm)
        sd["output.weight"].copy_(out_w)
    ids = torch.randint(0, V, (PROMPT + STEPS,), generator=torch.Generator().manual_seed(5))
    windows = {"w4096": None, "w128": 128}
    dtypes = {"bf16": BF, "fp32": torch.float32}
    # activations flowing between the one-layer pipeline stages: [variant][dtype] -> (prefill h, [decode h])
    acts = {v: {d: (None, [None] * STEPS) for d in dtypes} for v in windows}
    t0 = time.time()
    for l in range(L):
        w = _layer_weights(l, p, g)
        with torch.no_grad():
            for k, t in w.items():
                sd[k].copy_(t)
        extra = {}
        if l == 0:
            extra["tok_embeddings.weight"] = emb
        if l == L - 1:
            extra["norm.weight"] = final_norm
        for dn, dt in dtypes.items():
            wl = {k: t.to(dt) for k, t in {**w, **extra}.items()}
            om = mo.OracleModel(oargs, wl, pipeline_rank=l, num_pipeline_ranks=L)
            for vn, win in windows.items():
                oc = mo.OracleCache(1, 1, PROMPT + STEPS + 2, p["n_kv_heads"], p["head_dim"], win, dtype=dt)
                h_pre, h_dec = acts[vn][dn]
                h_pre = om.forward_partial(ids[:PROMPT], [PROMPT], oc, h_in=h_pre)
                h_dec = [om.forward_partial(ids[PROMPT + s:PROMPT + s + 1], [1], oc, h_in=h_dec[s]) for s in range(STEPS)]
                acts[vn][dn] = (h_pre, h_dec)
        del w
    oracle_s = time.time() - t0
    model._weights_changed()

    report = {}
    for vn, win in windows.items():
        cache = BufferCache(L, 1, PROMPT + STEPS + 2, p["n_kv_heads"], p["head_dim"], win, device="cuda", dtype=BF)
        cache.reset()
        with torch.inference_mode():
            hip = [model.forward(ids[:PROMPT].cuda(), [PROMPT], cache).cpu()]
            hip += [model.forward(ids[PROMPT + s:PROMPT + s + 1].cuda(), [1], cache).cpu() for s in range(STEPS)]
        hip = torch.cat(hip)                                              # [PROMPT + STEPS, V] fp32
        ref = {}
        for dn, dt in dtypes.items():
            h_pre, h_dec = acts[vn][dn]                                   # already RMS-normalised by the last stage
            ref[dn] = F.linear(torch.cat([h_pre] + h_dec), out_w.to(dt)).float()
        e_hip = (hip - ref["fp32"]).abs()
        e_o16 = (ref["bf16"] - ref["fp32"]).abs()
        d = (hip - ref["bf16"]).abs()
        report[vn] = dict(e_hip_max=float(e_hip.max()), e_o16_max=float(e_o16.max()), d_max=float(d.max()),
                          e_hip_mean=float(e_hip.mean()), e_o16_mean=float(e_o16.mean()), d_mean=float(d.mean()),
                          logit_absmax=float(ref["fp32"].abs().max()),
                          argmax_agree_hip=float((hip.argmax(1) == ref["fp32"].argmax(1)).float().mean()),
                          argmax_agree_o16=float((ref["bf16"].argmax(1) == ref["fp32"].argmax(1)).float().mean()))
    print(f"\n32-layer parity (oracle passes took {oracle_s:.0f} s on the host):")
    for vn, r in report.items():
        print(f"  {vn}: max|HIP-fp32| {r['e_hip_max']:.4f}  max|oracle_bf16-fp32| {r['e_o16_max']:.4f}  max|HIP-oracle_bf16| "
              f"{r['d_max']:.4f}   means {r['e_hip_mean']:.5f} / {r['e_o16_mean']:.5f} / {r['d_mean']:.5f}   |logit|max "
              f"{r['logit_absmax']:.2f}   argmax agreement with fp32: HIP {r['argmax_agree_hip']:.3f}, oracle_bf16 {r['argmax_agree_o16']:.3f}")
    from mistral_inference import _hip
    st = _hip.decode_engine_status(model._backend._workspace)
    assert st["engine_launches"] >= 2 * STEPS and st["status"] == 0, st   # the decode steps ran on the persistent engine
    for vn, r in report.items():
        assert r["e_hip_max"] <= 1.25 * r["e_o16_max"], (vn, r)
        assert r["e_hip_mean"] <= 1.10 * r["e_o16_mean"], (vn, r)
        assert r["argmax_agree_hip"] >= r["argmax_agree_o16"] - 0.02, (vn, r)
    del model
    torch.cuda.empty_cache()


def test_8_layers_4096_token_prompt_and_ring_wrap_vs_oracle():
    """The HEADLINE prefill shape against the oracle (round 2 only had self-consistency at T = 4096): 8 layers of the
    BASELINE configs[1] dims, one 4096-token prompt through the 256-query / 8-wave prefill attention and the 256x256 GEMMs,
    then 4 teacher-forced decode steps on the persistent engine that wrap the 4096-slot ring (positions 4096..4099
    overwrite slots 0..3).  Same three-number criterion as the 32-layer test, oracle layer-major in bf16 and fp32."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(P7B, n_layers=8, sliding_window=4096)
    L, V, D = p["n_layers"], p["vocab_size"], p["dim"]
    T, steps = 4096, 4
    oargs = mo.OracleArgs.from_params(p)
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda").eval()
    sd = dict(model.named_parameters())
    g = torch.Generator().manual_seed(4321)
    emb = torch.randn(V, D, generator=g).to(BF)
    final_norm = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
    out_w = _lin(V, D, g)
    with torch.no_grad():
        sd["tok_embeddings.weight"].copy_(emb)
        sd["norm.weight"].copy_(final_norm)
        sd["output.weight"].copy_(out_w)
    ids = torch.randint(0, V, (T + steps,), generator=torch.Generator().manual_seed(6))
    dtypes = {"bf16": BF, "fp32": torch.float32}
    acts = {d: (None, [None] * steps) for d in dtypes}
    t0 = time.time()
    for l in range(L):
        w = _layer_weights(l, p, g)
        with torch.no_grad():
            for k, t in w.items():
                sd[k].copy_(t)
        extra = {}
        if l == 0:
            extra["tok_embeddings.weight"] = emb
        if l == L - 1:
            extra["norm.weight"] = final_norm
        for dn, dt in dtypes.items():
            wl = {k: t.to(dt) for k, t in {**w, **extra}.items()}
            om = mo.OracleModel(oargs, wl, pipeline_rank=l, num_pipeline_ranks=L)
            oc = mo.OracleCache(1, 1, T + steps + 2, p["n_kv_heads"], p["head_dim"], 4096, dtype=dt)
            h_pre, h_dec = acts[dn]
            h_pre = om.forward_partial(ids[:T], [T], oc, h_in=h_pre)
            h_dec = [om.forward_partial(ids[T + s:T + s + 1], [1], oc, h_in=h_dec[s]) for s in range(steps)]
            acts[dn] = (h_pre, h_dec)
        del w
    oracle_s = time.time() - t0
    model._weights_changed()
    cache = BufferCache(L, 1, T + steps + 2, p["n_kv_heads"], p["head_dim"], 4096, device="cuda", dtype=BF)
    cache.reset()
    with torch.inference_mode():
        hip_pre = model.forward(ids[:T].cuda(), [T], cache)
        hip = [hip_pre[-260:].cpu(), hip_pre[::17].cpu()]   # the last rows (longest contexts) + a stride over all of them
        sel = torch.cat([torch.arange(T - 260, T), torch.arange(0, T, 17)])
        del hip_pre
        hip += [model.forward(ids[T + s:T + s + 1].cuda(), [1], cache).cpu() for s in range(steps)]
    hip = torch.cat(hip)
    ref = {}
    for dn, dt in dtypes.items():
        h_pre, h_dec = acts[dn]
        ref[dn] = F.linear(torch.cat([h_pre[sel]] + h_dec), out_w.to(dt)).float()
    e_hip, e_o16, d = (hip - ref["fp32"]).abs(), (ref["bf16"] - ref["fp32"]).abs(), (hip - ref["bf16"]).abs()
    n_dec = steps
    print(f"\n8 layers, 4096-token prompt + {steps} decode steps over the ring wrap (oracle: {oracle_s:.0f} s on the host): "
          f"max|HIP-fp32| {float(e_hip.max()):.4f}  max|oracle_bf16-fp32| {float(e_o16.max()):.4f}  max|HIP-oracle_bf16| "
          f"{float(d.max()):.4f}  means {float(e_hip.mean()):.5f} / {float(e_o16.mean()):.5f} / {float(d.mean()):.5f}  "
          f"decode rows only: max|HIP-fp32| {float(e_hip[-n_dec:].max()):.4f} vs {float(e_o16[-n_dec:].max()):.4f}  "
          f"|logit|max {float(ref['fp32'].abs().max()):.2f}")
    from mistral_inference import _hip
    st = _hip.decode_engine_status(model._backend._workspace)
    assert st["engine_launches"] >= steps and st["status"] == 0, st
    assert float(e_hip.max()) <= 1.25 * float(e_o16.max()), (float(e_hip.max()), float(e_o16.max()))
    assert float(e_hip.mean()) <= 1.10 * float(e_o16.mean())
    assert float(e_hip[-n_dec:].max()) <= 1.5 * float(e_o16.max())   # the wrapped-ring decode rows on their own
    agree_hip = float((hip.argmax(1) == ref["fp32"].argmax(1)).float().mean())
    agree_o16 = float((ref["bf16"].argmax(1) == ref["fp32"].argmax(1)).float().mean())
    assert agree_hip >= agree_o16 - 0.02, (agree_hip, agree_o16)
    del model
    torch.cuda.empty_cache()


def test_mixtral_8x7b_dims_4_layers_vs_oracle():
    """Inter-layer MoE accumulation at real width (round 2 had one layer per MoE config): 4 layers of the BASELINE
    configs[3] dims (8 experts top-2, 11.6 GB), 20-token prompt + 4 teacher-forced decode steps against the bf16 oracle
    (layer-major, tests/moe_depth_util.py).  The seed is one for which no router pick of the run is a near-tie (checked
    again here), so every row is compared."""
    import moe_depth_util as mu
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(mu.P8X7B_4L)
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda").eval()
    sd = dict(model.named_parameters())

    def sink(w):
        with torch.no_grad():
            for k, t in w.items():
                sd[k].copy_(t)

    ids, ref, gap = mu.oracle_run(sink=sink)
    assert gap > 2.5, f"router near-tie in the oracle run (gap {gap:.2f} ulp): pick another seed (python tests/moe_depth_util.py)"
    model._weights_changed()
    T, steps = mu.PROMPT, mu.STEPS
    cache = BufferCache(p["n_layers"], 1, T + steps + 2, 8, 128, None, device="cuda", dtype=BF)
    cache.reset()
    with torch.inference_mode():
        got = [model.forward(ids[:T].cuda(), [T], cache).cpu()]
        got += [model.forward(ids[T + i:T + i + 1].cuda(), [1], cache).cpu() for i in range(steps)]
    got = torch.cat(got)
    d = (got - ref).abs()
    print(f"\nMixtral-8x7B dims x 4 layers: max|HIP - oracle_bf16| {float(d.max()):.4f} (prefill rows {float(d[:T].max()):.4f}, "
          f"decode rows {float(d[T:].max()):.4f}), mean {float(d.mean()):.5f}, |logit|max {float(ref.abs().max()):.2f}, "
          f"min router gap {gap:.1f} ulp")
    assert float(d.max()) <= 4e-2 and float(d.mean()) <= 3e-3
    assert float((got.argmax(1) == ref.argmax(1)).float().mean()) >= 0.9
    del model
    torch.cuda.empty_cache()


def test_mixtral_8x7b_dims_one_layer_vs_oracle():
    """BASELINE.json configs[3] shapes (dim 4096, 32 q heads over 8 kv heads, hidden 14336, 8 experts top-2), ONE layer
    (2.9 GB), small vocabulary: prefill logits of a 48-token prompt and 4 teacher-forced decode steps against the bf16
    oracle; tokens whose router pick is a near-tie (bf16 logits within 2 ulp) are excluded, as for the 8x22B shapes."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(dim=4096, n_layers=1, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
             vocab_size=2048, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2))
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda")
    g = torch.Generator(device="cuda").manual_seed(17)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("norm.weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g, device="cuda"))
            elif name.startswith("tok_embeddings"):
                prm.copy_(torch.randn(prm.shape, generator=g, device="cuda"))
            else:
                prm.copy_((torch.rand(prm.shape, generator=g, device="cuda") * 2 - 1) / prm.shape[1] ** 0.5)
    model._weights_changed()
    model.eval()
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    oargs = mo.OracleArgs.from_params(p)
    T, steps = 48, 4
    ids = torch.randint(0, 2048, (T + steps,), generator=torch.Generator().manual_seed(18))
    cache = BufferCache(1, 1, T + steps + 2, 8, 128, None, device="cuda", dtype=BF)
    cache.reset()
    got = [model.forward(ids[:T].cuda(), [T], cache).cpu()]
    got += [model.forward(ids[T + i:T + i + 1].cuda(), [1], cache).cpu() for i in range(steps)]
    om = mo.OracleModel(oargs, w)
    oc = mo.OracleCache(1, 1, T + steps + 2, 8, 128, None, dtype=BF)
    mo.ROUTER_TRACE = []
    ref = [om.forward(ids[:T], [T], oc)] + [om.forward(ids[T + i:T + i + 1], [1], oc) for i in range(steps)]
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    kept = 0
    for gl, rl, lg in zip(got, ref, trace):
        srt = torch.sort(lg, dim=1, descending=True).values
        clear = (srt[:, 1] - srt[:, 2]) > 2 * srt[:, 1].abs().clamp(min=1e-3) * 2.0 ** -7
        kept += int(clear.sum())
        if clear.any():
            assert float((gl[clear] - rl[clear]).abs().max()) <= 4e-2
    assert kept >= 0.8 * (T + steps)


def test_nemo_12b_dims_4_layers_8300_token_chunked_prompt_vs_oracle():
    """BASELINE configs[2] beyond one layer (round 3 pinned Nemo by ONE layer at vocab 1024 and 300 tokens): 4 layers of the
    Mistral-Nemo-12B dims (dim 5120 != n_heads * 128 = 4096, hidden 14336) with the REAL vocabulary (131072: the LM-head
    GEMM at prefill, the LM-head GEMV / engine rows at decode), an 8300-token prompt fed in chunks of 2048 (no sliding window:
    later chunks attend to every cached key, the ring is 8306 slots - more than 8192) and 4 teacher-forced decode steps,
    against the layer-major oracle in bf16 AND fp32 (the three-number criterion of the 32-layer test).  The oracle's LM head
    is evaluated on a subset of the prompt rows (all rows of the last chunk, a stride over the rest) and on every decode row."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(dim=5120, n_layers=4, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
             vocab_size=131072, rope_theta=1e6)
    L, V, D = p["n_layers"], p["vocab_size"], p["dim"]
    T, steps, chunk = 8300, 4, 2048
    oargs = mo.OracleArgs.from_params(p)
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda").eval()
    sd = dict(model.named_parameters())
    g = torch.Generator().manual_seed(777)
    emb = torch.randn(V, D, generator=g).to(BF)
    final_norm = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
    out_w = _lin(V, D, g)
    with torch.no_grad():
        sd["tok_embeddings.weight"].copy_(emb)
        sd["norm.weight"].copy_(final_norm)
        sd["output.weight"].copy_(out_w)
    ids = torch.randint(0, V, (T + steps,), generator=torch.Generator().manual_seed(778))
    starts = list(range(0, T, chunk))
    dtypes = {"bf16": BF, "fp32": torch.float32}
    acts = {d: ([None] * len(starts), [None] * steps) for d in dtypes}
    t0 = time.time()
    for l in range(L):
        w = _layer_weights(l, p, g)
        with torch.no_grad():
            for k, t in w.items():
                sd[k].copy_(t)
        extra = {}
        if l == 0:
            extra["tok_embeddings.weight"] = emb
        if l == L - 1:
            extra["norm.weight"] = final_norm
        for dn, dt in dtypes.items():
            wl = {k: t.to(dt) for k, t in {**w, **extra}.items()}
            om = mo.OracleModel(oargs, wl, pipeline_rank=l, num_pipeline_ranks=L)
            oc = mo.OracleCache(1, 1, T + steps + 2, p["n_kv_heads"], p["head_dim"], None, dtype=dt)
            h_pre, h_dec = acts[dn]
            h_pre = [om.forward_partial(ids[s:min(s + chunk, T)], [min(s + chunk, T) - s], oc, h_in=h_pre[i])
                     for i, s in enumerate(starts)]
            h_dec = [om.forward_partial(ids[T + s:T + s + 1], [1], oc, h_in=h_dec[s]) for s in range(steps)]
            acts[dn] = (h_pre, h_dec)
        del w
    oracle_s = time.time() - t0
    model._weights_changed()
    cache = BufferCache(L, 1, T + steps + 2, p["n_kv_heads"], p["head_dim"], None, device="cuda", dtype=BF)
    cache.reset()
    sel = torch.cat([torch.arange(0, starts[-1], 29), torch.arange(starts[-1], T)])   # stride + the whole last chunk
    hip = []
    with torch.inference_mode():
        for s in starts:
            e = min(s + chunk, T)
            lg = model.forward(ids[s:e].cuda(), [e - s], cache)       # [e - s, 131072] fp32: the contractual forward()
            rows = sel[(sel >= s) & (sel < e)] - s
            hip.append(lg[rows.cuda()].cpu())
            del lg
        hip += [model.forward(ids[T + s:T + s + 1].cuda(), [1], cache).cpu() for s in range(steps)]
    hip = torch.cat(hip)
    ref = {}
    for dn, dt in dtypes.items():
        h_pre, h_dec = acts[dn]
        ref[dn] = F.linear(torch.cat([torch.cat(h_pre)[sel]] + h_dec), out_w.to(dt)).float()
    e_hip, e_o16, d = (hip - ref["fp32"]).abs(), (ref["bf16"] - ref["fp32"]).abs(), (hip - ref["bf16"]).abs()
    print(f"\nNemo-12B dims x 4 layers, V = 131072, {T}-token prompt in chunks of {chunk} + {steps} decode steps (oracle: "
          f"{oracle_s:.0f} s on the host): max|HIP-fp32| {float(e_hip.max()):.4f}  max|oracle_bf16-fp32| {float(e_o16.max()):.4f}  "
          f"max|HIP-oracle_bf16| {float(d.max()):.4f}  means {float(e_hip.mean()):.5f} / {float(e_o16.mean()):.5f} / "
          f"{float(d.mean()):.5f}  decode rows: max|HIP-fp32| {float(e_hip[-steps:].max()):.4f} vs {float(e_o16[-steps:].max()):.4f}  "
          f"|logit|max {float(ref['fp32'].abs().max()):.2f}")
    from mistral_inference import _hip
    st = _hip.decode_engine_status(model._backend._workspace)
    assert st["status"] == 0 and st["bad_id"] == 0, st
    assert float(e_hip.max()) <= 1.25 * float(e_o16.max()), (float(e_hip.max()), float(e_o16.max()))
    assert float(e_hip.mean()) <= 1.10 * float(e_o16.mean())
    assert float(e_hip[-steps:].max()) <= 1.5 * float(e_o16.max())
    agree_hip = float((hip.argmax(1) == ref["fp32"].argmax(1)).float().mean())
    agree_o16 = float((ref["bf16"].argmax(1) == ref["fp32"].argmax(1)).float().mean())
    assert agree_hip >= agree_o16 - 0.02, (agree_hip, agree_o16)
    # the fused prompt log-probabilities (GEMM_LOGPROB epilogue at V = 131072) against forward() + log_softmax on one chunk
    cache.reset()
    with torch.inference_mode():
        n = 1024
        tg = torch.cat([ids[1:n], torch.tensor([-1])]).to(torch.int32)
        lp, last = model.prompt_logprobs(ids[:n].cuda(), [n], cache, tg)
        cache.reset()
        full = torch.log_softmax(model.forward(ids[:n].cuda(), [n], cache), dim=-1)
        want = full[torch.arange(n - 1).cuda(), ids[1:n].cuda()]
        assert float((lp[:n - 1] - want).abs().max()) <= 2e-3
    del model
    torch.cuda.empty_cache()


def test_mixtral_8x22b_dims_3_layers_vs_oracle():
    """BASELINE configs[4] beyond one layer: 3 layers of the Mixtral-8x22B dims (dim 6144, 48 query heads over 8 kv heads =
    GQA ratio 6, hidden 16384, 8 experts top-2; 14.5 GB), 12-token prompt (the MFMA GEMM path, token-grouped MoE GEMM) + 3
    teacher-forced decode steps against the bf16 oracle, layer-major (tests/moe_depth_util.py; the seed is one whose run has
    no router near-tie, checked again here)."""
    import moe_depth_util as mu
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(mu.P8X22B_3L)
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda").eval()
    sd = dict(model.named_parameters())

    def sink(w):
        with torch.no_grad():
            for k, t in w.items():
                sd[k].copy_(t)

    T, steps = mu.PROMPT_22B, mu.STEPS_22B
    ids, ref, gap = mu.oracle_run(seed=mu.SEED_22B, sink=sink, p=p, prompt=T, steps=steps)
    # torch.topk's order on a near-tie is unspecified, and this host's bf16 matmul may round the router logits differently
    # from the build container's: rows from the first near-tie on (a picked-expert swap reaches every later token through
    # the K/V of the next layer) are not compared.  On the build container no row of this seed is near a tie.
    keep = mu.clean_rows(mu.oracle_run.token_gaps)
    assert int(keep.sum()) >= 5, (f"only {int(keep.sum())} rows of this host's oracle run are free of a router near-tie (closest call "
                                  f"{gap:.2f} ulp): pick another seed (python tests/moe_depth_util.py 0 8x22b all)")
    model._weights_changed()
    cache = BufferCache(p["n_layers"], 1, T + steps + 2, 8, 128, None, device="cuda", dtype=BF)
    cache.reset()
    with torch.inference_mode():
        got = [model.forward(ids[:T].cuda(), [T], cache).cpu()]
        got += [model.forward(ids[T + i:T + i + 1].cuda(), [1], cache).cpu() for i in range(steps)]
    got_all, ref_all = torch.cat(got), ref
    got, ref = got_all[keep], ref_all[keep]
    d = (got - ref).abs()
    print(f"\nMixtral-8x22B dims x 3 layers: max|HIP - oracle_bf16| {float(d.max()):.4f} over the {int(keep.sum())} of {T + steps} rows "
          f"free of a router near-tie ({int(keep[T:].sum())} of them decode rows), mean {float(d.mean()):.5f}; all rows: max "
          f"{float((got_all - ref_all).abs().max()):.4f}; |logit|max {float(ref_all.abs().max()):.2f}, closest router call {gap:.1f} ulp")
    assert float(d.max()) <= 4e-2 and float(d.mean()) <= 3e-3
    assert float((got.argmax(1) == ref.argmax(1)).float().mean()) >= 0.9
    del model
    torch.cuda.empty_cache()
