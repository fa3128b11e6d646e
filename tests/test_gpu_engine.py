"""Persistent decode engine (csrc/decode_engine.hip) against the launch path it replaces, stage by stage and end to end.

The engine reproduces the launch path's arithmetic AND summation orders (csrc/attn_decode_core.cuh, gemv_core.cuh), so
the contract is bit equality: logits, the residual stream and every K/V ring must be identical after every decode step.
The launch path itself is compared with the oracle in test_gpu_model.py / test_gpu_depth.py; one oracle comparison of
the engine is repeated here so that this file stands on its own.
"""
import os
import sys

import pytest
import torch

import mistral_oracle as mo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _model(args: mo.OracleArgs, seed: int):
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    w = mo.synth_weights(args, seed=seed)
    targs = TransformerArgs.from_dict(mo.params_json(args))
    targs.max_batch_size = 1
    with torch.device("meta"):
        m = Transformer(targs)
    m = m.to(BF).to_empty(device="cuda")
    m.load_state_dict({k: v.cuda() for k, v in w.items()}, assign=True)
    return m.eval(), w


def _cache(m, n):
    from mistral_inference.cache import BufferCache
    a = m.args
    c = BufferCache(m.n_local_layers, 1, n, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=BF)
    c.reset()
    return c


def _run(m, ids, prompt_len, steps, engine: bool, graph: bool = False):
    """Prefill `prompt_len` tokens (always the launch path), then `steps` teacher-forced decode steps."""
    from mistral_inference import _hip
    prev = _hip.set_decode_engine(engine)
    try:
        c = _cache(m, prompt_len + steps + 2)
        m.forward(ids[:prompt_len], [prompt_len], c)
        outs = []
        ctx = m.graphed_decode(c) if graph else _null()
        with ctx:
            for i in range(steps):
                outs.append(m.forward(ids[prompt_len + i:prompt_len + i + 1], [1], c)[0].clone())
        torch.cuda.synchronize()
        st = _hip.decode_engine_status(m._backend._workspace)
        # only slots that were written: the rings are torch.empty, so the rest is allocator garbage
        rings = []
        for l in range(m.n_local_layers):
            n = min(c.cache_sizes[l], prompt_len + steps)
            rings.append((c.cache_k[l][:, :n].clone(), c.cache_v[l][:, :n].clone()))
        return outs, rings, st
    finally:
        _hip.set_decode_engine(prev)


def _where(a, b):
    """Diagnostics for a ring mismatch: how many elements differ and the first few (slot, head, dim, ref, got)."""
    d = (a != b) | (torch.isnan(a.float()) != torch.isnan(b.float()))
    idx = d.nonzero()[:4].tolist()
    return int(d.sum()), [(i[1], i[2], i[3], float(a[tuple(i)]), float(b[tuple(i)])) for i in idx]


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


SHAPES = {
    # dims are multiples of 512 (engine requirement); vocab deliberately not a multiple of 512
    "gqa4_window_wraps": dict(dim=512, n_layers=3, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
                              vocab_size=1000, sliding_window=48),
    "mha_no_window": dict(dim=512, n_layers=2, head_dim=128, hidden_dim=1536, n_heads=4, n_kv_heads=4, norm_eps=1e-5,
                          vocab_size=514, sliding_window=None),
    "gqa2_long_ring": dict(dim=1024, n_layers=2, head_dim=128, hidden_dim=2048, n_heads=8, n_kv_heads=4, norm_eps=1e-6,
                           vocab_size=2048, sliding_window=1200),
    # per-layer window list (reference args.py:55-59 / cache.py:13-24): layer 0 keeps 16 slots, layer 1 the whole sequence
    "window_list": dict(dim=512, n_layers=2, head_dim=128, hidden_dim=1024, n_heads=4, n_kv_heads=1, norm_eps=1e-5,
                        vocab_size=512, sliding_window=[16, None]),
    # an 8.3K-slot ring: 31 splits of 272 slots = 136 K/V pieces per CU, more than the LDS ring holds at once
    "ring_longer_than_lds": dict(dim=512, n_layers=1, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
                                 vocab_size=640, sliding_window=None),
    # MoE on the engine (top-2 routing; router in every workgroup, the loader waits for its decision, two experts' W1|W3
    # and W2 slabs per layer): 8 experts as Mixtral, and 4 experts with MHA
    "moe_8_experts_top2": dict(dim=512, n_layers=3, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
                               vocab_size=1000, sliding_window=48, num_experts=8, num_experts_per_tok=2),
    "moe_4_experts_mha": dict(dim=1024, n_layers=2, head_dim=128, hidden_dim=1536, n_heads=8, n_kv_heads=8, norm_eps=1e-5,
                              vocab_size=514, sliding_window=None, num_experts=4, num_experts_per_tok=2),
    # MoE with 12 q|k|v units per workgroup (32 + 2 x 8 heads) and rows of 4 pieces: the holder waves of the MoE engine build
    # keep the last six units of every layer in registers, fetched during the previous layer's router bubble (ENG_QKV_HOLD)
    "moe_qkv_holders": dict(dim=2048, n_layers=3, head_dim=128, hidden_dim=2048, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                            vocab_size=1000, sliding_window=48, num_experts=4, num_experts_per_tok=2),
    # rows whose piece count is not a multiple of 4: dim 3072 = 6 pieces (streamed in 2-piece groups), hidden 1536 = 3 (single
    # pieces) - the Mistral-Nemo case (dim 5120 = 10 pieces) in small; 6 kv heads x 32 splits leave a quarter of the CUs without
    # attention work
    "rows_of_6_and_3_pieces": dict(dim=3072, n_layers=2, head_dim=128, hidden_dim=1536, n_heads=24, n_kv_heads=6, norm_eps=1e-5,
                                   vocab_size=768, sliding_window=64),
    # THE NEMO BUILD (round 6: decode_engine_nemo.o - dense GQA-4 models of a large dim whose rows are not multiples of 4 pieces,
    # every loader DMA from inline asm): Mistral-Nemo's rows (dim 5120 = 10 pieces, streamed as contiguous units; n_heads * 128
    # = 4096 = 8 pieces for Wo; hidden 3584 = 7 pieces for W2) at 2 layers, over a ring that wraps
    "nemo_rows_of_10_pieces": dict(dim=5120, n_layers=2, head_dim=128, hidden_dim=3584, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                                   vocab_size=1000, sliding_window=48),
    # ... and over a ring longer than the LDS ring holds at once (272-slot splits: the streamed K/V mode), contiguous K/V runs
    "nemo_rows_long_ring": dict(dim=5120, n_layers=1, head_dim=128, hidden_dim=2048, n_heads=16, n_kv_heads=4, norm_eps=1e-5,
                                vocab_size=640, sliding_window=None),
    # HOLDER WAVES at a size the whole suite can afford: they need dim % 2048 == 0 and >= 11 W1|W3 units per CU
    # (decode_engine.hip holder_units): hidden_dim 5632 = 11 units x 256 CUs exactly
    "holders_mid_size": dict(dim=2048, n_layers=2, head_dim=128, hidden_dim=5632, n_heads=16, n_kv_heads=4, norm_eps=1e-5,
                             vocab_size=1024, sliding_window=256),
    # more layers than one launch takes (ENG_MAXL = 32): two engine launches per step, the residual stream handed over
    # through global memory, the LM head and the step's commit only in the second (BASELINE configs[2], Nemo: 40 layers)
    "two_launches_34_layers": dict(dim=512, n_layers=34, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2,
                                   norm_eps=1e-5, vocab_size=600, sliding_window=40),
}


def test_two_launch_depth_vs_oracle_and_graph():
    """34 layers = 32 + 2: engine == launch path bit for bit is in test_engine_bit_equal_launch_path; here graph replay
    of the two-launch step and the CPU oracle (the path BASELINE configs[2] runs on had no parity evidence)."""
    p = SHAPES["two_launches_34_layers"]
    args = mo.OracleArgs(**p)
    m, w = _model(args, seed=13)
    prompt_len, steps = 30, 14   # crosses the 40-slot ring
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(6)).cuda()
    ref, _, st0 = _run(m, ids, prompt_len, steps, engine=False)
    eager, _, st1 = _run(m, ids, prompt_len, steps, engine=True)
    graph, _, st2 = _run(m, ids, prompt_len, steps, engine=True, graph=True)
    assert st1["status"] == 0 and st2["status"] == 0
    assert st1["engine_launches"] - st0["engine_launches"] == 2 * steps   # two launches per step
    assert all(torch.equal(a, b) for a, b in zip(ref, eager))
    assert all(torch.equal(a, b) for a, b in zip(ref, graph))
    om = mo.OracleModel(args, w)
    oc = mo.OracleCache(args.n_layers, 1, prompt_len + steps + 2, args.n_kv_heads, args.head_dim, args.sliding_window, dtype=BF)
    om.forward(ids[:prompt_len].cpu(), [prompt_len], oc)
    worst = 0.0
    for i in range(steps):
        o = om.forward(ids[prompt_len + i:prompt_len + i + 1].cpu(), [1], oc)[0]
        worst = max(worst, float((eager[i].cpu() - o).abs().max()))
    assert worst < 6e-2, worst   # 34 layers of bf16 storage (the 32-layer floor of test_gpu_depth is 4.7e-2)


@pytest.mark.parametrize("holders", [1, 0])
def test_holder_waves_fingerprint(holders):
    """Holder waves reduce their W1|W3 unit from REGISTERS, not from the ring: same arithmetic, same order - made visible
    the way test_engine_summation_order_fingerprint does it for q|k|v.  W1/W3[:, D/2:] = -W1/W3[:, :D/2] and ffn_norm
    symmetric: with an input whose halves are equal every hidden value would be the rounding residue of the summation
    order.  h1 = h + Wo(attn) is not symmetric, so the property is arranged one level down: Wo rows are duplicated
    (row D/2 + i = row i) and the embedding halves equal, which makes h1's halves equal.  The residues then flow through
    W2 into layer 1's K/V rows and the logits: any difference in order shows up as a bit difference against the launch
    path, with the holder waves on and off."""
    from mistral_inference import _hip
    p = SHAPES["holders_mid_size"]
    args = mo.OracleArgs(**p)
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    w = mo.synth_weights(args, seed=23)
    half = p["dim"] // 2
    w["tok_embeddings.weight"][:, half:] = w["tok_embeddings.weight"][:, :half]
    for l in range(p["n_layers"]):
        w[f"layers.{l}.attention.wo.weight"][half:] = w[f"layers.{l}.attention.wo.weight"][:half]
        w[f"layers.{l}.ffn_norm.weight"][half:] = w[f"layers.{l}.ffn_norm.weight"][:half]
        w[f"layers.{l}.feed_forward.w2.weight"][half:] = w[f"layers.{l}.feed_forward.w2.weight"][:half]
        # silu(residue) * residue ~ 1e-8: scaled by an exact power of two so that it survives the bf16 residual add and
        # steers everything downstream (layer 1's K/V rows, the logits)
        w[f"layers.{l}.feed_forward.w2.weight"] *= 2.0 ** 24
        for n in ("w1", "w3"):
            t = w[f"layers.{l}.feed_forward.{n}.weight"]
            t[:, half:] = -t[:, :half]
    targs = TransformerArgs.from_dict(mo.params_json(args))
    targs.max_batch_size = 1
    with torch.device("meta"):
        m = Transformer(targs)
    m = m.to(BF).to_empty(device="cuda")
    m.load_state_dict({k: v.cuda() for k, v in w.items()}, assign=True)
    m.eval()
    prompt_len, steps = 7, 6
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(4)).cuda()
    _hip.check(_hip.lib().mi_debug_set_engine_holders(holders), "holders")
    try:
        ref, ref_rings, _ = _run(m, ids, prompt_len, steps, engine=False)
        got, got_rings, st = _run(m, ids, prompt_len, steps, engine=True)
    finally:
        _hip.check(_hip.lib().mi_debug_set_engine_holders(-1), "holders")
    assert st["status"] == 0 and st["engine_launches"] > 0
    assert all(torch.isfinite(a).all() for a in ref)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (holders, i, float((a - b).abs().max()))
    for l, ((k0, v0), (k1, v1)) in enumerate(zip(ref_rings, got_rings)):
        assert torch.equal(k0, k1) and torch.equal(v0, v1), (holders, l, _where(k0, k1), _where(v0, v1))


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_engine_bit_equal_launch_path(name):
    p = SHAPES[name]
    args = mo.OracleArgs(**p)
    m, _ = _model(args, seed=11)
    W = p["sliding_window"] if isinstance(p["sliding_window"], int) else 10 ** 9
    prompt_len = 40 if W < 100 else 300  # 40 + steps crosses the 48-slot ring; 300 leaves later splits empty in a 1200 ring
    steps = 12
    if name == "rows_of_6_and_3_pieces":
        prompt_len = 57  # + 12 steps crosses the 64-slot ring
    if name == "two_launches_34_layers":
        prompt_len = 33  # + 12 steps crosses the 40-slot ring
    if name in ("ring_longer_than_lds", "nemo_rows_long_ring"):
        prompt_len, steps = 8290, 6
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(3)).cuda()
    from mistral_inference import _hip
    # (the `nemo` build is opt-in - it measured slower than the launch path at the Nemo-12B dims -: engine variant 3 routes to it)
    prev = _hip.lib().mi_debug_set_engine_variant(3) if name.startswith("nemo_") else None
    try:
        ref, ref_rings, st0 = _run(m, ids, prompt_len, steps, engine=False)
        got, got_rings, st1 = _run(m, ids, prompt_len, steps, engine=True)
    finally:
        if prev is not None:
            _hip.lib().mi_debug_set_engine_variant(prev)
    assert st1["status"] == 0 and st1["abort"] == 0, st1
    assert st1["engine_launches"] >= steps and st0["engine_launches"] < st1["engine_launches"]  # the engine really ran
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.isfinite(b).all(), i
        assert torch.equal(a, b), (name, i, float((a - b).abs().max()))
    for l, ((k0, v0), (k1, v1)) in enumerate(zip(ref_rings, got_rings)):
        assert torch.equal(k0, k1) and torch.equal(v0, v1), (name, l, _where(k0, k1), _where(v0, v1))


# The WIDE build of the engine (csrc/decode_engine.hip compiled with -DENG_WIDE=1: GQA ratio 4 and 6, a 7-fill ring, rows that
# are not a multiple of 4 pieces streamed as contiguous units).  mi_debug_set_engine_variant(1) makes it the first choice, so
# its code paths run at sizes the suite can afford; the first three shapes are declined by the shipped build anyway.
WIDE_SHAPES = {
    # Mixtral-8x22B in small: GQA ratio 6 (48 / 8 heads), every row a multiple of 4 pieces, top-2 MoE
    "gqa6_moe_rows_of_4": dict(dim=2048, n_layers=2, head_dim=128, hidden_dim=2048, n_heads=48, n_kv_heads=8, norm_eps=1e-5,
                               vocab_size=1000, sliding_window=48, num_experts=4, num_experts_per_tok=2),
    # Mixtral-8x22B's rows: dim 6144 = 12 pieces, GQA ratio 6, MoE: the holder waves keep ONE q|k|v unit each of the next layer
    # (run_qkv_holder1: three 4-piece groups per row)
    "gqa6_moe_rows_of_12": dict(dim=6144, n_layers=2, head_dim=128, hidden_dim=2048, n_heads=48, n_kv_heads=8, norm_eps=1e-5,
                                vocab_size=1000, sliding_window=48, num_experts=4, num_experts_per_tok=2),
    # Mistral-Nemo's rows: dim 5120 = 10 pieces (contiguous units of 20 and 40 pieces), n_heads * 128 != dim
    "rows_of_10_pieces": dict(dim=5120, n_layers=2, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
                              vocab_size=1002, sliding_window=48),
    # GQA ratio 6 dense, odd piece counts everywhere (dim 2560 = 5, Wo rows 1536 = 3, hidden 1536 = 3): the unaligned tails of
    # a contiguous unit go piece by piece
    "gqa6_rows_of_5_and_3": dict(dim=2560, n_layers=2, head_dim=128, hidden_dim=1536, n_heads=12, n_kv_heads=2, norm_eps=1e-5,
                                 vocab_size=770, sliding_window=None),
    # shapes the shipped build takes as well, forced through the wide one: the 7-fill ring under the shipped row layout,
    # a ring longer than the LDS ring (K/V pieces streamed with per-piece bookkeeping), MoE at GQA ratio 4
    "gqa4_window_wraps": SHAPES["gqa4_window_wraps"],
    "ring_longer_than_lds": SHAPES["ring_longer_than_lds"],
    "moe_8_experts_top2": SHAPES["moe_8_experts_top2"],
    "rows_of_6_and_3_pieces": dict(SHAPES["rows_of_6_and_3_pieces"], n_heads=24, n_kv_heads=6),
}


@pytest.mark.parametrize("name", sorted(WIDE_SHAPES))
def test_wide_engine_build_bit_equal_launch_path(name):
    from mistral_inference import _hip
    p = WIDE_SHAPES[name]
    args = mo.OracleArgs(**p)
    m, _ = _model(args, seed=17)
    W = p["sliding_window"] if isinstance(p["sliding_window"], int) else 10 ** 9
    prompt_len, steps = (40, 12) if W < 100 else (300, 8)
    if name == "ring_longer_than_lds":
        prompt_len, steps = 8290, 6
    if name == "rows_of_6_and_3_pieces":
        prompt_len = 57
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(3)).cuda()
    prev = _hip.lib().mi_debug_set_engine_variant(1)
    try:
        ref, ref_rings, st0 = _run(m, ids, prompt_len, steps, engine=False)
        got, got_rings, st1 = _run(m, ids, prompt_len, steps, engine=True)
        graph, _, st2 = _run(m, ids, prompt_len, steps, engine=True, graph=True)
    finally:
        _hip.lib().mi_debug_set_engine_variant(prev)
    assert st1["status"] == 0 and st1["abort"] == 0 and st2["status"] == 0, (st1, st2)
    assert st1["engine_launches"] - st0["engine_launches"] >= steps   # an engine build really ran (for the first three: the wide one)
    for i, (a, b, c) in enumerate(zip(ref, got, graph)):
        assert torch.isfinite(b).all(), i
        assert torch.equal(a, b), (name, i, float((a - b).abs().max()))
        assert torch.equal(a, c), (name, "graph", i)
    for l, ((k0, v0), (k1, v1)) in enumerate(zip(ref_rings, got_rings)):
        assert torch.equal(k0, k1) and torch.equal(v0, v1), (name, l, _where(k0, k1), _where(v0, v1))


# The NEXT build (decode_engine_next.o = the same source under build_native.ENGINE_NEXT_FLAGS: abort word read on every 1024th spin,
# consumer waves at s_setprio 1, holders fetch from the K/V stage on, every loader DMA from inline asm in the SGPR-base form, no
# stamp sites, the loader not stopped during the hid sweep): the dense GQA-4 shapes whose rows are all multiples of 4 pieces -
# the headline model - are routed to it.  None of SHAPES qualifies, so its code runs here at sizes the
# suite can afford, against the launch path AND against the frozen default object (mi_debug_set_engine_variant(2)).
NEXT_SHAPES = {
    # holder waves on (12 W1|W3 units per CU), ring wraps
    "holders_window_wraps": dict(dim=2048, n_layers=3, head_dim=128, hidden_dim=6144, n_heads=16, n_kv_heads=4, norm_eps=1e-5,
                                 vocab_size=1000, sliding_window=48),
    # no holder waves (4 units per CU), 1200-slot ring with empty later splits, some CUs without a W2 / Wo unit pair
    "no_holders_long_ring": dict(dim=2048, n_layers=2, head_dim=128, hidden_dim=2048, n_heads=16, n_kv_heads=4, norm_eps=1e-6,
                                 vocab_size=514, sliding_window=1200),
    # the headline's widths at 2 layers: dim 4096, 32 / 8 heads, hidden 14336
    "headline_widths": dict(dim=4096, n_layers=2, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                            vocab_size=2048, sliding_window=64),
}


@pytest.mark.parametrize("name", sorted(NEXT_SHAPES))
def test_next_engine_build_bit_equal_launch_path_and_frozen_build(name):
    from mistral_inference import _hip
    p = NEXT_SHAPES[name]
    m, _ = _model(mo.OracleArgs(**p), seed=19)
    prompt_len, steps = (40, 14) if p["sliding_window"] < 100 else (300, 8)
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(3)).cuda()
    ref, ref_rings, st0 = _run(m, ids, prompt_len, steps, engine=False)
    got, got_rings, st1 = _run(m, ids, prompt_len, steps, engine=True)
    graph, _, st2 = _run(m, ids, prompt_len, steps, engine=True, graph=True)
    prev = _hip.lib().mi_debug_set_engine_variant(2)
    try:
        frozen, frozen_rings, st3 = _run(m, ids, prompt_len, steps, engine=True)
    finally:
        _hip.lib().mi_debug_set_engine_variant(prev)
    assert st1["status"] == 0 and st1["abort"] == 0 and st2["status"] == 0 and st3["status"] == 0, (st1, st2, st3)
    assert st1["engine_launches"] - st0["engine_launches"] >= steps and st3["engine_launches"] - st2["engine_launches"] >= steps
    for i, (a, b, c, d) in enumerate(zip(ref, got, graph, frozen)):
        assert torch.isfinite(b).all(), i
        assert torch.equal(a, b), (name, i, float((a - b).abs().max()))
        assert torch.equal(a, c), (name, "graph", i)
        assert torch.equal(a, d), (name, "frozen", i)
    for l, ((k0, v0), (k1, v1), (k2, v2)) in enumerate(zip(ref_rings, got_rings, frozen_rings)):
        assert torch.equal(k0, k1) and torch.equal(v0, v1), (name, l, _where(k0, k1), _where(v0, v1))
        assert torch.equal(k0, k2) and torch.equal(v0, v2), (name, "frozen", l)


def test_engine_right_after_a_one_token_prompt():
    """kv_len = 2, 3, ...: every split but the first is empty, the current slot is the only other key."""
    p = SHAPES["gqa4_window_wraps"]
    m, _ = _model(mo.OracleArgs(**p), seed=2)
    ids = torch.randint(0, p["vocab_size"], (6,), generator=torch.Generator().manual_seed(8)).cuda()
    ref, ref_rings, _ = _run(m, ids, 1, 5, engine=False)
    got, got_rings, st = _run(m, ids, 1, 5, engine=True)
    assert st["status"] == 0 and st["engine_launches"] >= 5
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    assert all(torch.equal(k0, k1) and torch.equal(v0, v1) for (k0, v0), (k1, v1) in zip(ref_rings, got_rings))


def test_engine_summation_order_fingerprint():
    """fp32 summation ORDER of the engine's row dots == the launch path's, made visible: with W[:, D/2:] = -W[:, :D/2]
    and an input whose two halves are equal, every q/k/v output of layer 0 is mathematically zero - what is computed is
    the rounding residue of the particular association order, a fingerprint of it that survives the bf16 rounding."""
    p = dict(dim=4096, n_layers=1, head_dim=128, hidden_dim=1024, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
             vocab_size=512, sliding_window=64)
    args = mo.OracleArgs(**p)
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    w = mo.synth_weights(args, seed=21)
    half = p["dim"] // 2
    w["tok_embeddings.weight"][:, half:] = w["tok_embeddings.weight"][:, :half]
    w["layers.0.attention_norm.weight"][half:] = w["layers.0.attention_norm.weight"][:half]
    for n in ("wq", "wk", "wv"):
        t = w[f"layers.0.attention.{n}.weight"]
        t[:, half:] = -t[:, :half]
    targs = TransformerArgs.from_dict(mo.params_json(args))
    targs.max_batch_size = 1
    with torch.device("meta"):
        m = Transformer(targs)
    m = m.to(BF).to_empty(device="cuda")
    m.load_state_dict({k: v.cuda() for k, v in w.items()}, assign=True)
    m.eval()
    prompt_len, steps = 5, 6
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(4)).cuda()
    _, ref_rings, _ = _run(m, ids, prompt_len, steps, engine=False)
    _, got_rings, st = _run(m, ids, prompt_len, steps, engine=True)
    assert st["status"] == 0
    (k0, v0), (k1, v1) = ref_rings[0], got_rings[0]
    dec_k, dec_v = k0[:, prompt_len:], v0[:, prompt_len:]           # rows written by the decode steps
    assert float(dec_k.float().abs().max()) < 1e-3 and float(dec_v.float().abs().max()) < 1e-3   # residues, not signal
    assert float((dec_v != 0).float().mean()) > 0.5                # ... and not trivially zero: the fingerprint is real
    assert torch.equal(k0, k1) and torch.equal(v0, v1), (_where(k0, k1), _where(v0, v1))


def test_engine_graph_replay_and_oracle():
    """hipGraph replay of engine steps (the epoch lives in device memory, so replays see fresh tags) == eager engine
    steps == launch path; and the whole thing against the CPU oracle."""
    p = SHAPES["gqa4_window_wraps"]
    args = mo.OracleArgs(**p)
    m, w = _model(args, seed=5)
    prompt_len, steps = 30, 24
    ids = torch.randint(0, p["vocab_size"], (prompt_len + steps,), generator=torch.Generator().manual_seed(9)).cuda()
    ref, _, _ = _run(m, ids, prompt_len, steps, engine=False)
    eager, _, st = _run(m, ids, prompt_len, steps, engine=True)
    graph, _, st2 = _run(m, ids, prompt_len, steps, engine=True, graph=True)
    assert st["status"] == 0 and st2["status"] == 0
    assert all(torch.equal(a, b) for a, b in zip(ref, eager))
    assert all(torch.equal(a, b) for a, b in zip(ref, graph))
    om = mo.OracleModel(args, w)
    oc = mo.OracleCache(args.n_layers, 1, prompt_len + steps + 2, args.n_kv_heads, args.head_dim, args.sliding_window, dtype=BF)
    om.forward(ids[:prompt_len].cpu(), [prompt_len], oc)
    for i in range(steps):
        o = om.forward(ids[prompt_len + i:prompt_len + i + 1].cpu(), [1], oc)[0]
        assert float((eager[i].cpu() - o).abs().max()) < 4e-2, i


def test_engine_full_size_bit_equal():
    """BASELINE configs[1] dims, 32 layers, 4096-token prompt, ring full and wrapping: engine == launch path bit for bit
    (logits of 6 decode steps, all 64 rings), eager and replayed from a hipGraph."""
    sys.path.insert(0, ROOT)
    import bench
    m = bench.build_model(dict(bench.MISTRAL_7B), 0, 1, "cuda")
    T, steps = 4096, 6
    ids = torch.randint(0, m.args.vocab_size, (T + steps,), generator=torch.Generator().manual_seed(0)).cuda()
    ref, ref_rings, _ = _run(m, ids, T, steps, engine=False)
    got, got_rings, st = _run(m, ids, T, steps, engine=True)
    assert st["status"] == 0 and st["abort"] == 0, st
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    for l, ((k0, v0), (k1, v1)) in enumerate(zip(ref_rings, got_rings)):
        assert torch.equal(k0, k1) and torch.equal(v0, v1), (l, _where(k0, k1), _where(v0, v1))
    del ref_rings, got_rings
    graph, _, st2 = _run(m, ids, T, steps, engine=True, graph=True)
    assert st2["status"] == 0
    assert all(torch.equal(a, b) for a, b in zip(ref, graph))
    del m
    torch.cuda.empty_cache()


def test_engine_soak_greedy_1200_tokens_equal_launch_path():
    """The granule hand-offs are "observed untorn", not an architectural guarantee (MI355X_MICROARCH.md), and a rare stale or
    torn word would not crash - it would change a number.  1200 consecutive greedy tokens of the full 32-layer model at context
    4096+ (1200 steps x 32 layers x 7 edges x 256 workgroups of hand-offs), once on the engine and once on the launch path,
    must be the SAME tokens with the same log-probabilities: one wrong hand-off anywhere flips the sequence from there on."""
    sys.path.insert(0, ROOT)
    import bench
    from mistral_inference import _hip
    m = bench.build_model(dict(bench.MISTRAL_7B), 0, 1, "cuda")
    T, steps = 4096, 1200
    ids = torch.randint(0, m.args.vocab_size, (T,), generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for engine in (True, False):
        prev = _hip.set_decode_engine(engine)
        try:
            c = _cache(m, T + steps + 8)
            last = m.forward(ids, [T], c)[-1:]
            sess = m.greedy_session(c, torch.argmax(last, dim=-1))
            toks, lps = [], []
            left = steps
            while left:
                n = min(left, sess.HIST)
                sess.run(n)
                t, l = sess.collect()
                toks.append(t.clone())
                lps.append(l.clone())
                left -= n
            st = _hip.decode_engine_status(m._backend._workspace)
            assert st["status"] == 0, st
            outs.append((torch.cat(toks), torch.cat(lps), st["engine_launches"]))
        finally:
            _hip.set_decode_engine(prev)
    (t1, l1, n1), (t2, l2, n2) = outs
    assert n1 >= steps  # (the first session really ran on the engine)
    same = (t1 == t2).all(dim=1)
    first_diff = int((~same).nonzero()[0, 0]) if not bool(same.all()) else -1
    assert first_diff < 0, f"token sequences diverge at step {first_diff}"
    assert float((l1 - l2).abs().max()) < 1e-4
    del m
    torch.cuda.empty_cache()
