"""K/V ring layouts (ABI v7, include/mistral_hip.h MI_KV_SLOT_MAJOR / MI_KV_HEAD_MAJOR): the head-major storage that `BufferCache`
allocates is a LAYOUT decision only - every entry point that touches a ring gives the same bits in both layouts, and Python sees
the reference's shape [max_batch, W, n_kv_heads, head_dim] (cache.py:163-167) either way."""
import pytest
import torch

import mistral_oracle as mo
from test_gpu_engine import SHAPES, _model, _where

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _hip():
    from mistral_inference import _hip
    return _hip


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def dev_ring(t: torch.Tensor, head_major: bool) -> torch.Tensor:
    """Host ring [B, W, Hkv, Dh] -> device ring of the same logical content in the requested layout."""
    if head_major:
        return t.permute(0, 2, 1, 3).contiguous().cuda().permute(0, 2, 1, 3)
    return t.contiguous().cuda()


@pytest.mark.parametrize("H,Hkv", [(8, 2), (32, 8), (4, 4), (12, 2)])
@pytest.mark.parametrize("W,lens", [(16, [5, 16, 40]), (300, [1, 299, 300]), (4096, [4096, 17, 5000])])
def test_attn_decode_both_layouts(H, Hkv, W, lens):
    h = _hip()
    Dh, B = 128, len(lens)
    ck, cv = rnd(B, W, Hkv, Dh, seed=1), rnd(B, W, Hkv, Dh, seed=2)
    q = rnd(B, H * Dh, seed=3).cuda()
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32).cuda()
    outs = []
    for hm in (False, True):
        k, v = dev_ring(ck, hm), dev_ring(cv, hm)
        assert h.kv_layout_of(k) == (h.KV_HEAD_MAJOR if hm else h.KV_SLOT_MAJOR)
        outs.append(h.attn_decode(q, k, v, H, pos).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("W,seen,new", [(8, [4, 8, 21], [4, 4, 2]), (64, [100], [200]), (512, [600, 30, 0], [700, 520, 300]),
                                        (4096, [4096], [2048])])
def test_attn_prefill_both_layouts(W, seen, new):
    h = _hip()
    H, Hkv, Dh, B = 8, 2, 128, len(new)
    T = sum(new)
    ck, cv = rnd(B, W, Hkv, Dh, seed=4), rnd(B, W, Hkv, Dh, seed=5)
    qkv = rnd(T, (H + 2 * Hkv) * Dh, seed=6).cuda()
    q_start = torch.tensor([0] + list(torch.tensor(new).cumsum(0)), dtype=torch.int32).cuda()
    kv_before = torch.tensor(seen, dtype=torch.int32).cuda()
    outs = [h.attn_prefill(qkv, H, Hkv, Dh, dev_ring(ck, hm), dev_ring(cv, hm), W, q_start, kv_before, B, max(new)).cpu()
            for hm in (False, True)]
    assert torch.equal(outs[0], outs[1])


def test_ring_writes_both_layouts():
    """mi_kv_write (prefill) and the ring write of mi_qkv_rope_kvwrite (decode): the same logical ring content."""
    h = _hip()
    import mistral_oracle as mo
    B, W, Hkv, Dh, H, D = 3, 8, 2, 128, 4, 512
    new = [6, 3, 12]
    T = sum(new)
    k, v = rnd(T, Hkv * Dh, seed=7).cuda(), rnd(T, Hkv * Dh, seed=8).cuda()
    tok_seq = torch.tensor([b for b, n in enumerate(new) for _ in range(n)], dtype=torch.int32).cuda()
    tok_pos = torch.tensor([5 + i for i in range(6)] + [2 + i for i in range(3)] + list(range(12)), dtype=torch.int32).cuda()
    q_start = torch.tensor([0, 6, 9, 21], dtype=torch.int32).cuda()
    base_k, base_v = torch.full((B, W, Hkv, Dh), 7.0, dtype=BF), torch.full((B, W, Hkv, Dh), 9.0, dtype=BF)
    res = []
    for hm in (False, True):
        dk, dv = dev_ring(base_k, hm), dev_ring(base_v, hm)
        h.kv_write(dk, dv, k, v, tok_seq, tok_pos, q_start)
        res.append((dk.cpu(), dv.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not torch.equal(res[0][0], base_k)
    # decode-sized fused projection + ring write
    Td = 3
    x = rnd(Td, D, seed=9, scale=2.0).cuda()
    wq, wk, wv = (rnd(n, D, seed=10 + i, scale=D ** -0.5).cuda() for i, n in enumerate((H * Dh, Hkv * Dh, Hkv * Dh)))
    cs = mo.rope_angles(Dh, 100, 1e6).cuda()
    pos = torch.tensor([3, 50, 9], dtype=torch.int32).cuda()
    seq = torch.tensor([2, 0, 1], dtype=torch.int32).cuda()
    res = []
    for hm in (False, True):
        dk, dv = dev_ring(base_k, hm), dev_ring(base_v, hm)
        out = h.qkv_rope_kvwrite(x, wq, wk, wv, Dh, cs, pos, cache_k=dk, cache_v=dv, tok_seq=seq).cpu()
        res.append((out, dk.cpu(), dv.cpu()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1][2, 3 % W].reshape(-1), res[0][0][0, H * Dh:(H + Hkv) * Dh])  # row of sequence 2, slot 3


@pytest.mark.parametrize("shape", ["gqa4_window_wraps", "mha_no_window", "gqa2_long_ring", "ring_longer_than_lds", "moe_8_experts_top2"])
@pytest.mark.parametrize("engine", [True, False])
def test_model_bit_equal_in_both_layouts(shape, engine, monkeypatch):
    """A chunked prefill and 70 decode steps (engine / launch path) with the cache in the reference's layout (MI_KV_LAYOUT=0) and
    in the head-major layout: logits of every step and the rings' logical content are equal bit for bit - the engine's loader
    streams head-major K/V pieces as 4-KiB runs in another ring order, the arithmetic order does not change."""
    from mistral_inference import _hip
    from mistral_inference.cache import BufferCache
    m, _ = _model(mo.OracleArgs(**SHAPES[shape]), seed=23)
    a = m.args
    ids = torch.randint(0, a.vocab_size, (200,), generator=torch.Generator().manual_seed(3)).cuda()
    prompt, steps = 90, 70
    prev = _hip.set_decode_engine(engine)
    try:
        results = []
        for lay in ("0", "1"):
            monkeypatch.setenv("MI_KV_LAYOUT", lay)
            c = BufferCache(m.n_local_layers, 1, prompt + steps + 2, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=BF)
            assert c.kv_layout == int(lay) and tuple(c.cache_k[0].shape[2:]) == (a.n_kv_heads, a.head_dim)
            c.reset()
            outs = [m.forward(ids[:60], [60], c).clone(), m.forward(ids[60:prompt], [prompt - 60], c).clone()]
            for i in range(steps):
                outs.append(m.forward(ids[prompt + i:prompt + i + 1], [1], c)[0].clone())
            torch.cuda.synchronize()
            st = _hip.decode_engine_status(m._backend._workspace)
            assert st["status"] == 0, st
            rings = []
            for l in range(m.n_local_layers):
                n = min(c.cache_sizes[l], prompt + steps)
                rings.append((c.cache_k[l][:, :n].contiguous(), c.cache_v[l][:, :n].contiguous()))
            results.append((outs, rings))
        for x, y in zip(results[0][0], results[1][0]):
            assert torch.equal(x, y)
        for (k0, v0), (k1, v1) in zip(results[0][1], results[1][1]):
            assert torch.equal(k0, k1), _where(k0, k1)
            assert torch.equal(v0, v1), _where(v0, v1)
    finally:
        _hip.set_decode_engine(prev)
