"""Seeded random-shape sweep of the leaf operators against fp32 torch math on the same bf16 inputs: shapes nobody picked by
hand (odd token counts, ragged batches, windows around tile edges, K/N that are not tile multiples)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from hip_util import bf16_ulp_close

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _hip():
    from mistral_inference import _hip as h
    return h


def _rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(BF)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_linear(seed):
    h = _hip()
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    M = rng.choice([1, 2, 7, 8, 9, 31, 100, 255, 256, 257, 513, 1000, 1500])
    K = 8 * rng.randint(2, 260)
    n_seg = rng.randint(1, 3)
    rows = [8 * rng.randint(1, 90) for _ in range(n_seg)]
    x = _rnd(g, M, K)
    ws = [_rnd(g, n, K, scale=1 / math.sqrt(K)) for n in rows]
    ref = torch.cat([F.linear(x.float(), w.float()) for w in ws], dim=1)
    got = h.linear(x.cuda(), tuple(w.cuda() for w in ws), h.EPI_STORE).cpu()
    ok, err = bf16_ulp_close(got, ref.to(BF), ulps=1.0)
    assert ok, (M, K, rows, err)
    N = rows[0]
    w3 = _rnd(g, N, K, scale=1 / math.sqrt(K))
    a, b = F.linear(x, ws[0]), F.linear(x, w3)
    r3 = (F.silu(a) * b).float()
    g3 = h.linear(x.cuda(), (ws[0].cuda(), w3.cuda()), h.EPI_SWIGLU).cpu().float()
    e3 = (g3 - r3).abs()
    assert bool((e3 <= 4 * r3.abs() * 2.0 ** -7 + 8e-3).all()), (M, K, N, float(e3.max()))


def _attn_ref(q, k, v, qpos, kpos, W):
    """q [s, H, Dh], k/v [n, Hkv, Dh]; visible iff qp - W < kp <= qp."""
    H, Hkv = q.shape[1], k.shape[1]
    R = H // Hkv
    kk, vv = k.float().repeat_interleave(R, 1), v.float().repeat_interleave(R, 1)
    s = torch.einsum("shd,nhd->hsn", q.float(), kk) / math.sqrt(q.shape[-1])
    vis = (kpos[None, :] <= qpos[:, None]) & (kpos[None, :] > qpos[:, None] - W)
    s = s.masked_fill(~vis[None], float("-inf"))
    return torch.einsum("hsn,nhd->shd", torch.softmax(s, -1), vv)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_attention(seed):
    h = _hip()
    rng = random.Random(100 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    H, Hkv = rng.choice([(4, 2), (8, 8), (12, 2), (32, 8), (16, 2), (4, 4)])
    Dh = 128
    W = rng.choice([5, 33, 64, 100, 129, 300, 1000])
    B = rng.randint(1, 3)
    seen = [rng.choice([0, 0, 3, W - 1, W, W + 7, 2 * W + 1]) for _ in range(B)]
    new = [rng.randint(1, 3 * 128 + 5) for _ in range(B)]
    T = sum(new)
    ck = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    cv = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    hist = []
    for b in range(B):
        kh, vh = _rnd(g, seen[b], Hkv, Dh), _rnd(g, seen[b], Hkv, Dh)
        for p in range(seen[b]):
            ck[b, p % W], cv[b, p % W] = kh[p], vh[p]
        hist.append((kh, vh))
    qkv = _rnd(g, T, (H + 2 * Hkv) * Dh)
    q_start = torch.tensor([0] + list(torch.tensor(new).cumsum(0)), dtype=torch.int32)
    got = h.attn_prefill(qkv.cuda(), H, Hkv, Dh, ck.cuda(), cv.cuda(), W, q_start.cuda(),
                         torch.tensor(seen, dtype=torch.int32).cuda(), B, max(new)).cpu()
    nq, nkv = H * Dh, Hkv * Dh
    o = 0
    for b, s in enumerate(new):
        p = seen[b]
        rows = qkv[o:o + s]
        n_old = min(p, W)
        keys = torch.cat([hist[b][0][p - n_old:p], rows[:, nq:nq + nkv].reshape(s, Hkv, Dh)])
        vals = torch.cat([hist[b][1][p - n_old:p], rows[:, nq + nkv:].reshape(s, Hkv, Dh)])
        ref = _attn_ref(rows[:, :nq].reshape(s, H, Dh), keys, vals, torch.arange(p, p + s), torch.arange(p - n_old, p + s), W)
        err = float((got[o:o + s].float().view(s, H, Dh) - ref).abs().max())
        assert err <= 2.5e-2, (seed, H, Hkv, W, seen, new, b, err)
        o += s
    # decode step on top: one more token per sequence, ring updated by hand
    lens = [seen[b] + new[b] + 1 for b in range(B)]
    o = 0
    allk, allv = [], []
    for b, s in enumerate(new):
        rows = qkv[o:o + s]
        allk.append(torch.cat([hist[b][0], rows[:, nq:nq + nkv].reshape(s, Hkv, Dh), _rnd(g, 1, Hkv, Dh)]))
        allv.append(torch.cat([hist[b][1], rows[:, nq + nkv:].reshape(s, Hkv, Dh), _rnd(g, 1, Hkv, Dh)]))
        o += s
    ck2 = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    cv2 = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    for b in range(B):
        for p in range(max(0, lens[b] - W), lens[b]):
            ck2[b, p % W], cv2[b, p % W] = allk[b][p], allv[b][p]
    qd = _rnd(g, B, H * Dh)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32)
    gd = h.attn_decode(qd.cuda(), ck2.cuda(), cv2.cuda(), H, pos.cuda()).cpu()
    for b in range(B):
        lo = max(0, lens[b] - W)
        ref = _attn_ref(qd[b].view(1, H, Dh), allk[b][lo:], allv[b][lo:], torch.tensor([lens[b] - 1]),
                        torch.arange(lo, lens[b]), W)
        ok, err = bf16_ulp_close(gd[b].view(1, H, Dh), ref, ulps=2.0, floor=1e-2)
        assert ok, (seed, "decode", H, Hkv, W, lens, b, err)
