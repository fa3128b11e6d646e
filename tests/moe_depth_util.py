"""Layer-major oracle run of a multi-layer Mixtral-8x7B-dims model (tests/test_gpu_depth.py::test_mixtral_8x7b_dims_4_layers).

Weights are generated on the HOST layer by layer from a seeded CPU generator (bit-reproducible: same torch build here and on
the GPU box), handed to `sink(name -> tensor)` (the GPU test copies them into the HIP model), used by a one-layer oracle
stage and dropped: 2.9 GB per layer, never 11.6 GB at once.  The router's bf16 logits of every (layer, forward) are recorded so
that the caller can tell whether any top-2 pick was a near-tie (torch.topk's tie order is unspecified)."""
import math

import torch
import torch.nn.functional as F

import mistral_oracle as mo

BF = torch.bfloat16
P8X7B_4L = dict(dim=4096, n_layers=4, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                vocab_size=2048, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2))
PROMPT, STEPS = 20, 4
SEED = 17  # chosen with `python tests/moe_depth_util.py`: the closest router call of this run is 2.87 bf16 ulp (most seeds: < 1)
# BASELINE configs[4] dims (GQA ratio 6, dim 6144, hidden 16384), 3 layers = 14.5 GB: `python tests/moe_depth_util.py 0 8x22b`
P8X22B_3L = dict(dim=6144, n_layers=3, head_dim=128, hidden_dim=16384, n_heads=48, n_kv_heads=8, norm_eps=1e-5,
                 vocab_size=2048, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2))
PROMPT_22B, STEPS_22B = 12, 3
SEED_22B = 12  # of seeds 0-21 the one with the widest margin on a GPU box's host (closest router call 7.3 bf16 ulp there; host
# arithmetic differs in the last bit between machines - seed 9: 3.66 ulp on the build container, 1.84 on the box): the test
# additionally drops the rows behind any near-tie of ITS oracle run (clean_rows)


def _lin(o, i, g):
    return ((torch.rand(o, i, generator=g) * 2 - 1) / math.sqrt(i)).to(BF)


def layer_weights(l, p, g):
    D, Fh, nq, nkv = p["dim"], p["hidden_dim"], p["n_heads"] * p["head_dim"], p["n_kv_heads"] * p["head_dim"]
    pre = f"layers.{l}."
    w = {
        pre + "attention.wq.weight": _lin(nq, D, g), pre + "attention.wk.weight": _lin(nkv, D, g),
        pre + "attention.wv.weight": _lin(nkv, D, g), pre + "attention.wo.weight": _lin(D, nq, g),
        pre + "attention_norm.weight": (1 + 0.1 * torch.randn(D, generator=g)).to(BF),
        pre + "ffn_norm.weight": (1 + 0.1 * torch.randn(D, generator=g)).to(BF),
        pre + "feed_forward.gate.weight": _lin(p["moe"]["num_experts"], D, g),
    }
    for e in range(p["moe"]["num_experts"]):
        w[pre + f"feed_forward.experts.{e}.w1.weight"] = _lin(Fh, D, g)
        w[pre + f"feed_forward.experts.{e}.w2.weight"] = _lin(D, Fh, g)
        w[pre + f"feed_forward.experts.{e}.w3.weight"] = _lin(Fh, D, g)
    return w


def oracle_run(seed=SEED, sink=None, p=P8X7B_4L, prompt=PROMPT, steps=STEPS):
    """Returns (ids, logits [prompt + steps, V] fp32 of the bf16 oracle, min relative gap between the 2nd and 3rd router
    logit in units of a bf16 ulp, over every (layer, token))."""
    L, V, D = p["n_layers"], p["vocab_size"], p["dim"]
    oargs = mo.OracleArgs.from_params(p)
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(V, D, generator=g).to(BF)
    final_norm = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
    out_w = _lin(V, D, g)
    if sink:
        sink({"tok_embeddings.weight": emb, "norm.weight": final_norm, "output.weight": out_w})
    ids = torch.randint(0, V, (prompt + steps,), generator=torch.Generator().manual_seed(seed + 100))
    h_pre, h_dec = None, [None] * steps
    min_gap = float("inf")
    tok_gap = torch.full((L, prompt + steps), float("inf"))  # [layer, token (prompt rows, then decode rows)]
    for l in range(L):
        w = layer_weights(l, p, g)
        if sink:
            sink(w)
        extra = {}
        if l == 0:
            extra["tok_embeddings.weight"] = emb
        if l == L - 1:
            extra["norm.weight"] = final_norm
        om = mo.OracleModel(oargs, {**w, **extra}, pipeline_rank=l, num_pipeline_ranks=L)
        oc = mo.OracleCache(1, 1, prompt + steps + 2, p["n_kv_heads"], p["head_dim"], None, dtype=BF)
        mo.ROUTER_TRACE = []
        h_pre = om.forward_partial(ids[:prompt], [prompt], oc, h_in=h_pre)
        h_dec = [om.forward_partial(ids[prompt + s:prompt + s + 1], [1], oc, h_in=h_dec[s]) for s in range(steps)]
        trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
        row = 0
        for lg in trace:  # one entry per forward call of this layer: the prompt, then each decode step
            srt = torch.sort(lg, dim=1, descending=True).values
            ulp = srt[:, 1].abs().clamp(min=1e-3) * 2.0 ** -7
            gaps = (srt[:, 1] - srt[:, 2]) / ulp
            min_gap = min(min_gap, float(gaps.min()))
            tok_gap[l, row:row + gaps.numel()] = gaps
            row += gaps.numel()
        del w, om
    logits = F.linear(torch.cat([h_pre] + h_dec), out_w).float()
    oracle_run.token_gaps = tok_gap
    return ids, logits, min_gap


def clean_rows(token_gaps: torch.Tensor, thr: float = 2.5) -> torch.Tensor:
    """Rows whose logits cannot depend on a near-tie of the router (torch.topk's order on one is unspecified, and two hosts
    round the bf16 router logits differently in the last bit).  A near-tie of token t in layer l touches t's own row; unless l
    is the last layer it also reaches every LATER token, through t's K/V rows in layer l + 1."""
    L, N = token_gaps.shape
    bad = torch.zeros(N, dtype=torch.bool)
    for l in range(L):
        for t in (token_gaps[l] <= thr).nonzero().flatten().tolist():
            bad[t] = True
            if l < L - 1:
                bad[t + 1:] = True
    return ~bad


if __name__ == "__main__":  # seed search (host only): the first seed whose run has no near-tie (gap > 2.5 ulp everywhere)
    import sys
    import time
    big = len(sys.argv) > 2 and sys.argv[2] == "8x22b"
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, 64):
        t0 = time.time()
        _, lg, gap = oracle_run(seed, p=P8X22B_3L, prompt=PROMPT_22B, steps=STEPS_22B) if big else oracle_run(seed)
        n_clean = int(clean_rows(oracle_run.token_gaps).sum())
        print(f"seed {seed}: min (2nd - 3rd) router gap = {gap:.2f} bf16 ulp, rows free of a near-tie: {n_clean} of {lg.shape[0]}, "
              f"|logit|max {float(lg.abs().max()):.2f}, {time.time() - t0:.0f} s", flush=True)
        if gap > 2.5 and not (len(sys.argv) > 3 and sys.argv[3] == "all"):
            break
        if len(sys.argv) > 4 and seed + 1 >= int(sys.argv[4]):
            break
