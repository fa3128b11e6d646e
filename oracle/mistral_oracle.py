"""CPU oracle for the `Transformer.forward_partial` hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A functional restatement (plain torch on CPU, no nn.Module, no xformers) of the algorithm the
reference runs per forward.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this file; the shipped package never does and has no CPU path.

Parity status: PINNED.  `oracle/make_golden.py` runs the UNMODIFIED reference
(/root/reference/src, through `oracle/shim`) on seeded inputs and stores its outputs under
`tests/golden/`; `tests/test_oracle_golden.py` holds this file to those vectors.  The reference
itself ships no golden vector for the transformer path (its one KAT, tests/test_generate.py:196,
is Mamba); its two self-consistency tests (tests/test_generate.py:36-69, :199-230) are restated in
`tests/test_oracle_selfconsistency.py`.

Every rounding point follows the reference's execution in the storage dtype `dt` (bf16 for all
BASELINE configs): see SURVEY.md Appendix A.  Each function cites the reference lines it follows
(paths relative to /root/reference/src/mistral_inference/).

Differences in *formulation* (not results): attention visibility is computed from absolute token
positions -- key position kp is visible to query position qp of the same sequence iff
qp - W < kp <= qp -- which is what the three xformers masks built at cache.py:236-254 reduce to
(SURVEY.md Appendix B); GQA indexes kv_head = q_head // repeats instead of materialising
`repeat_interleave` (transformer_layers.py:16-19,84).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

ROPE_TABLE_LEN = 128_000  # transformer.py:116

# Tests may set this to a list to record every router call's bf16 logits [T, E] (as fp32), in call order.
# Used to tell a genuine mismatch from a top-k near-tie (torch.topk's tie order is unspecified and a 1-ulp
# difference in a bf16 logit can swap the k-th and (k+1)-th expert; SURVEY.md section 7 "hard parts").
ROUTER_TRACE: Optional[list] = None


@dataclass
class OracleArgs:
    """Subset of TransformerArgs (args.py:29-59) the hot path reads."""

    dim: int
    n_layers: int
    head_dim: int
    hidden_dim: int
    n_heads: int
    n_kv_heads: int
    norm_eps: float
    vocab_size: int
    rope_theta: Optional[float] = None
    num_experts: int = 0           # moe.num_experts (moe.py:11-13)
    num_experts_per_tok: int = 0   # moe.num_experts_per_tok
    sliding_window: Union[None, int, List[Optional[int]]] = None

    @staticmethod
    def from_params(p: dict) -> "OracleArgs":
        moe = p.get("moe") or {}
        sw = p.get("sliding_window", None)
        if sw is None:
            sw = p.get("_sliding_window", None)  # args.py:55-59
        return OracleArgs(
            dim=p["dim"], n_layers=p["n_layers"], head_dim=p["head_dim"], hidden_dim=p["hidden_dim"],
            n_heads=p["n_heads"], n_kv_heads=p["n_kv_heads"], norm_eps=p["norm_eps"],
            vocab_size=p["vocab_size"], rope_theta=p.get("rope_theta"),
            num_experts=moe.get("num_experts", 0), num_experts_per_tok=moe.get("num_experts_per_tok", 0),
            sliding_window=sw,
        )


# --------------------------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """transformer_layers.py:115-120: fp32 normalise -> round to x.dtype -> multiply by weight."""
    xf = x.float()
    inv = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (xf * inv).type_as(x) * weight


def rope_angles(head_dim: int, end: int, theta: float) -> torch.Tensor:
    """rope.py:6-10: returns fp32 [end, head_dim/2, 2] = (cos, sin) of pos * theta^(-2i/head_dim).

    Bitwise the real view of the reference's complex64 `freqs_cis` table (torch.polar(1, angle))."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(end), inv_freq).float()
    return torch.view_as_real(torch.polar(torch.ones_like(ang), ang)).contiguous()


def apply_rope(x: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """rope.py:13-23 for one tensor.  x [T, heads, Dh] storage dtype; cs [T, Dh/2, 2] fp32.

    Adjacent pairs (x[2i], x[2i+1]) rotate by the angle of slot i; products and the
    sum/difference are individually rounded fp32 operations (complex64 multiply), result is
    rounded back to x.dtype."""
    xf = x.float().reshape(*x.shape[:-1], -1, 2)
    a, b = xf[..., 0], xf[..., 1]
    c, s = cs[:, None, :, 0], cs[:, None, :, 1]
    re = a * c - b * s
    im = a * s + b * c
    return torch.stack((re, im), dim=-1).flatten(-2).type_as(x)


def swiglu_ffn(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """transformer_layers.py:105-106: w2(silu(w1 x) * w3 x); every intermediate is in x.dtype."""
    return F.linear(F.silu(F.linear(x, w1)) * F.linear(x, w3), w2)


def moe_ffn(x: torch.Tensor, gate: torch.Tensor, experts: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
            top_k: int) -> torch.Tensor:
    """moe.py:24-32.  Router logits in x.dtype, top-k on them, fp32 softmax over the k picked
    logits rounded to x.dtype, then experts visited in ascending id, each adding
    round(w * y_e) into a zero-initialised x.dtype accumulator."""
    logits = F.linear(x, gate)
    if ROUTER_TRACE is not None:
        ROUTER_TRACE.append(logits.float().clone())
    top_w, top_i = torch.topk(logits, top_k)
    top_w = torch.softmax(top_w, dim=1, dtype=torch.float).to(x.dtype)
    out = torch.zeros_like(x)
    for e, (w1, w2, w3) in enumerate(experts):
        tok, slot = torch.where(top_i == e)
        if tok.numel() == 0:
            continue
        out[tok] += top_w[tok, slot, None] * swiglu_ffn(x[tok], w1, w2, w3)
    return out


# --------------------------------------------------------------------------------------------
# rotating K/V buffer (cache.py:13-24, 140-195)
# --------------------------------------------------------------------------------------------
def layer_cache_sizes(n_layers: int, max_seq_len: int, sliding_window) -> List[int]:
    """cache.py:13-24."""
    if sliding_window is None:
        return [max_seq_len] * n_layers
    if isinstance(sliding_window, int):
        return [sliding_window] * n_layers
    assert n_layers % len(sliding_window) == 0
    pat = [w if w is not None else max_seq_len for w in sliding_window]
    return pat * (n_layers // len(sliding_window))


@dataclass
class OracleCache:
    """Per-layer rings K,V [max_batch, W_layer, Hkv, Dh] plus tokens-seen per sequence."""

    n_layers: int
    max_batch_size: int
    max_seq_len: int
    n_kv_heads: int
    head_dim: int
    sliding_window: Union[None, int, List[Optional[int]]] = None
    dtype: torch.dtype = torch.float32
    sizes: List[int] = field(init=False)
    k: List[torch.Tensor] = field(init=False)
    v: List[torch.Tensor] = field(init=False)
    seen: Optional[List[int]] = field(init=False, default=None)

    def __post_init__(self) -> None:
        self.sizes = layer_cache_sizes(self.n_layers, self.max_seq_len, self.sliding_window)
        shape = lambda w: (self.max_batch_size, w, self.n_kv_heads, self.head_dim)  # noqa: E731
        self.k = [torch.zeros(shape(w), dtype=self.dtype) for w in self.sizes]
        self.v = [torch.zeros(shape(w), dtype=self.dtype) for w in self.sizes]

    def reset(self) -> None:
        self.seen = None


def _attend(q: torch.Tensor, keys: torch.Tensor, vals: torch.Tensor, qpos: torch.Tensor, kpos: torch.Tensor,
            window: Optional[int], causal: bool) -> torch.Tensor:
    """softmax(q k^T / sqrt(Dh) + mask) v in fp32 (transformer_layers.py:87-88 + xformers).

    q [s, H, Dh]; keys/vals [n, Hkv, Dh]; qpos [s], kpos [n] absolute positions."""
    s, H, Dh = q.shape
    Hkv = keys.shape[1]
    rep = H // Hkv
    qf = q.float().view(s, Hkv, rep, Dh)
    kf, vf = keys.float(), vals.float()
    scores = torch.einsum("sgrd,ngd->grsn", qf, kf) * (Dh ** -0.5)
    if causal:
        vis = kpos[None, :] <= qpos[:, None]
        if window is not None:
            vis &= kpos[None, :] > qpos[:, None] - window
        scores = scores.masked_fill(~vis[None, None], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    out = torch.einsum("grsn,ngd->sgrd", p, vf)
    return out.reshape(s, H * Dh).to(q.dtype)


def attention_block(x: torch.Tensor, wq, wk, wv, wo, cs: torch.Tensor, args: OracleArgs, seqlens: List[int],
                    cache: Optional[OracleCache], layer: int) -> torch.Tensor:
    """transformer_layers.py:56-93 with the cache branches of :72-81 and cache.py:83-117,226-259.

    x is the (already normalised) input [T, D]; returns wo(attn) [T, D]."""
    T = x.shape[0]
    H, Hkv, Dh = args.n_heads, args.n_kv_heads, args.head_dim
    q = apply_rope(F.linear(x, wq).view(T, H, Dh), cs)
    k = apply_rope(F.linear(x, wk).view(T, Hkv, Dh), cs)
    v = F.linear(x, wv).view(T, Hkv, Dh)

    if cache is None:
        # transformer_layers.py:72-73 + :165 (mask never forwarded): every token sees every token.
        pos = torch.arange(T)
        o = _attend(q, k, v, pos, pos, None, causal=False)
        return F.linear(o, wo)

    W = cache.sizes[layer]
    ring_k, ring_v = cache.k[layer], cache.v[layer]
    outs = []
    start = 0
    for b, s in enumerate(seqlens):
        p = cache.seen[b]
        qb, kb, vb = q[start:start + s], k[start:start + s], v[start:start + s]
        n_old = min(p, W)
        old_pos = torch.arange(p - n_old, p)
        old_k = ring_k[b, old_pos % W]           # cache.py:59-67 (unrotate) expressed by position
        old_v = ring_v[b, old_pos % W]
        keys = torch.cat([old_k, kb]) if n_old else kb
        vals = torch.cat([old_v, vb]) if n_old else vb
        kpos = torch.cat([old_pos, torch.arange(p, p + s)])
        qpos = torch.arange(p, p + s)
        outs.append(_attend(qb, keys, vals, qpos, kpos, W, causal=True))
        # ring write, cache.py:83-92 with to_cache_mask / cache_positions of cache.py:226-235
        keep = torch.arange(s) >= s - W
        slots = (qpos % W)[keep]
        ring_k[b, slots] = kb[keep]
        ring_v[b, slots] = vb[keep]
        start += s
    return F.linear(torch.cat(outs), wo)


# --------------------------------------------------------------------------------------------
# model-level
# --------------------------------------------------------------------------------------------
def pipeline_layer_range(n_layers: int, rank: int, world: int) -> range:
    """transformer.py:94-97."""
    per = math.ceil(n_layers / world)
    return range(rank * per, min(n_layers, rank * per + per))


class OracleModel:
    """Weights as a flat dict with the checkpoint key names (checkpoint row)."""

    def __init__(self, args: OracleArgs, weights: Dict[str, torch.Tensor], pipeline_rank: int = 0,
                 num_pipeline_ranks: int = 1):
        self.args = args
        self.w = weights
        self.pipeline_rank = pipeline_rank
        self.num_pipeline_ranks = num_pipeline_ranks
        self.layer_ids = list(pipeline_layer_range(args.n_layers, pipeline_rank, num_pipeline_ranks))
        self.n_local_layers = len(self.layer_ids)
        self._cs: Optional[torch.Tensor] = None

    @property
    def dtype(self) -> torch.dtype:
        return next(iter(self.w.values())).dtype

    @property
    def angles(self) -> torch.Tensor:
        if self._cs is None:
            self._cs = rope_angles(self.args.head_dim, ROPE_TABLE_LEN, self.args.rope_theta or 1000000.0)
        return self._cs

    def _ffn(self, i: int, x: torch.Tensor) -> torch.Tensor:
        a, w = self.args, self.w
        pre = f"layers.{i}.feed_forward."
        if a.num_experts:
            experts = [(w[f"{pre}experts.{e}.w1.weight"], w[f"{pre}experts.{e}.w2.weight"],
                        w[f"{pre}experts.{e}.w3.weight"]) for e in range(a.num_experts)]
            return moe_ffn(x, w[pre + "gate.weight"], experts, a.num_experts_per_tok)
        return swiglu_ffn(x, w[pre + "w1.weight"], w[pre + "w2.weight"], w[pre + "w3.weight"])

    def block(self, i: int, local_i: int, h: torch.Tensor, cs, seqlens, cache) -> torch.Tensor:
        """transformer_layers.py:158-169."""
        a, w = self.args, self.w
        pre = f"layers.{i}."
        r = attention_block(rms_norm(h, w[pre + "attention_norm.weight"], a.norm_eps),
                            w[pre + "attention.wq.weight"], w[pre + "attention.wk.weight"],
                            w[pre + "attention.wv.weight"], w[pre + "attention.wo.weight"],
                            cs, a, seqlens, cache, local_i)
        h = h + r
        r = self._ffn(i, rms_norm(h, w[pre + "ffn_norm.weight"], a.norm_eps))
        return h + r

    def forward_partial(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[OracleCache] = None,
                        h_in: Optional[torch.Tensor] = None, collect: Optional[list] = None) -> torch.Tensor:
        """transformer.py:163-219.  `h_in` stands for the tensor a non-zero rank would `recv`."""
        assert sum(seqlens) == input_ids.shape[0]
        if cache is not None:
            if cache.seen is None:
                cache.seen = [0] * len(seqlens)
            assert len(cache.seen) == len(seqlens), "did you forget to reset cache?"
            starts = cache.seen
        else:
            starts = [0] * len(seqlens)
        positions = torch.cat([torch.arange(p, p + s) for p, s in zip(starts, seqlens)])
        cs = self.angles[positions]
        if self.pipeline_rank == 0 and h_in is None:
            h = F.embedding(input_ids, self.w["tok_embeddings.weight"])
        else:  # received from the previous rank - or, on rank 0, the multimodal embeddings of transformer.py:190-191
            assert h_in is not None
            h = h_in
        for local_i, i in enumerate(self.layer_ids):
            h = self.block(i, local_i, h, cs, seqlens, cache)
            if collect is not None:
                collect.append(h.clone())
        if cache is not None:
            cache.seen = [p + s for p, s in zip(cache.seen, seqlens)]
        if self.pipeline_rank < self.num_pipeline_ranks - 1:
            return h
        return rms_norm(h, self.w["norm.weight"], self.args.norm_eps)

    def forward(self, input_ids, seqlens, cache=None, h_in=None) -> torch.Tensor:
        """transformer.py:221-242 on the last rank (single-rank use): LM head then `.float()`."""
        h = self.forward_partial(input_ids, seqlens, cache, h_in)
        return F.linear(h, self.w["output.weight"]).float()


def generate(prompts: List[List[int]], model: OracleModel, *, max_tokens: int, max_batch_size: Optional[int] = None,
             chunk_size: Optional[int] = None, eos_id: Optional[int] = None
             ) -> Tuple[List[List[int]], List[List[float]]]:
    """Greedy (temperature 0) restatement of generate.py:43-148."""
    a = model.args
    B = len(prompts)
    lens = [len(p) for p in prompts]
    cache = OracleCache(model.n_local_layers, max_batch_size or B, max(lens) + max_tokens, a.n_kv_heads,
                        a.head_dim, a.sliding_window, dtype=model.dtype)
    logprobs: List[List[float]] = [[] for _ in range(B)]
    last = None
    chunk = chunk_size or max(lens)
    for s in range(0, max(lens), chunk):
        parts = [p[s:s + chunk] for p in prompts]
        assert all(len(p) > 0 for p in parts)
        pre = model.forward(torch.tensor(sum(parts, []), dtype=torch.long), [len(p) for p in parts], cache)
        lsm = torch.log_softmax(pre, dim=-1)
        if last is not None:
            prev = torch.log_softmax(last, dim=-1)
            for b in range(B):
                logprobs[b].append(prev[b, parts[b][0]].item())
        off = 0
        for b, part in enumerate(parts):
            logprobs[b].extend(lsm[off + i, part[i + 1]].item() for i in range(len(part) - 1))
            off += len(part)
        ends = torch.tensor([len(p) for p in parts]).cumsum(0) - 1
        last = pre.index_select(0, ends)
    out: List[torch.Tensor] = []
    done = torch.zeros(B, dtype=torch.bool)
    for _ in range(max_tokens):
        nxt = torch.argmax(last, dim=-1)
        if eos_id is not None:
            done |= nxt == eos_id
        if done.all():
            break
        lsm = torch.log_softmax(last, dim=-1)
        for b in range(B):
            logprobs[b].append(lsm[b, nxt[b]].item())
        out.append(nxt[:, None])
        last = model.forward(nxt, [1] * B, cache)
    toks = torch.cat(out, 1).tolist() if out else []
    return toks, logprobs


# --------------------------------------------------------------------------------------------
# synthetic checkpoints (the reference tests' own init, tests/test_generate.py:37-51)
# --------------------------------------------------------------------------------------------
def top_p_distribution(logits: torch.Tensor, temperature: float, p: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """What the reference's `sample` hands to torch.multinomial (generate.py:151-167), for ONE fp32 logits row [V]:
    (order [V] long, kept [V] float64) - `order` = token ids by descending probability of softmax(logits / temperature)
    (ties by ascending id: a STABLE sort; torch.sort leaves tie order unspecified), `kept` = the sorted probabilities with
    every position whose mass BEFORE it exceeds p zeroed, renormalised to sum 1.  The arithmetic follows the reference in
    fp32 (softmax, cumsum, the `probs_sum - probs_sort > p` mask) and only the final renormalisation is widened."""
    probs = torch.softmax(logits.float() / temperature, dim=-1)
    probs_sort, order = torch.sort(probs, dim=-1, descending=True, stable=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    mask = probs_sum - probs_sort > p
    kept = probs_sort.double().masked_fill(mask, 0.0)
    return order, kept / kept.sum()


def top_p_inverse_cdf(order: torch.Tensor, kept: torch.Tensor, u: float) -> int:
    """The token an inverse-CDF draw with the uniform variate u in [0, 1) picks from `top_p_distribution`'s output: the first
    sorted position whose cumulative kept mass exceeds u (torch.multinomial draws from the same distribution with its own
    stream, generate.py:168-169)."""
    cdf = torch.cumsum(kept, dim=-1)
    pos = int(torch.searchsorted(cdf, torch.tensor(u, dtype=cdf.dtype), right=True))
    last = int((kept > 0).nonzero()[-1])
    return int(order[min(pos, last)])


def synth_weights(args: OracleArgs, seed: int = 42, dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """nn.Linear-style U(+-1/sqrt(fan_in)), N(0,1) embeddings, norm weights slightly off 1 so the
    weight multiply is exercised; generated per tensor from a seeded generator, then cast."""
    g = torch.Generator().manual_seed(seed)

    def lin(o: int, i: int) -> torch.Tensor:
        b = 1.0 / math.sqrt(i)
        return ((torch.rand(o, i, generator=g) * 2 - 1) * b).to(dtype)

    def nrm(n: int) -> torch.Tensor:
        return (1.0 + 0.1 * torch.randn(n, generator=g)).to(dtype)

    a = args
    w: Dict[str, torch.Tensor] = {"tok_embeddings.weight": torch.randn(a.vocab_size, a.dim, generator=g).to(dtype)}
    for i in range(a.n_layers):
        p = f"layers.{i}."
        w[p + "attention.wq.weight"] = lin(a.n_heads * a.head_dim, a.dim)
        w[p + "attention.wk.weight"] = lin(a.n_kv_heads * a.head_dim, a.dim)
        w[p + "attention.wv.weight"] = lin(a.n_kv_heads * a.head_dim, a.dim)
        w[p + "attention.wo.weight"] = lin(a.dim, a.n_heads * a.head_dim)
        w[p + "attention_norm.weight"] = nrm(a.dim)
        w[p + "ffn_norm.weight"] = nrm(a.dim)
        if a.num_experts:
            w[p + "feed_forward.gate.weight"] = lin(a.num_experts, a.dim)
            for e in range(a.num_experts):
                q = f"{p}feed_forward.experts.{e}."
                w[q + "w1.weight"] = lin(a.hidden_dim, a.dim)
                w[q + "w2.weight"] = lin(a.dim, a.hidden_dim)
                w[q + "w3.weight"] = lin(a.hidden_dim, a.dim)
        else:
            w[p + "feed_forward.w1.weight"] = lin(a.hidden_dim, a.dim)
            w[p + "feed_forward.w2.weight"] = lin(a.dim, a.hidden_dim)
            w[p + "feed_forward.w3.weight"] = lin(a.hidden_dim, a.dim)
    w["norm.weight"] = nrm(a.dim)
    w["output.weight"] = lin(a.vocab_size, a.dim)
    return w


def params_json(args: OracleArgs) -> dict:
    p = {k: getattr(args, k) for k in ("dim", "n_layers", "head_dim", "hidden_dim", "n_heads", "n_kv_heads",
                                       "norm_eps", "vocab_size")}
    if args.rope_theta is not None:
        p["rope_theta"] = args.rope_theta
    if args.num_experts:
        p["moe"] = {"num_experts": args.num_experts, "num_experts_per_tok": args.num_experts_per_tok}
    if args.sliding_window is not None:
        p["sliding_window"] = args.sliding_window
    return p
