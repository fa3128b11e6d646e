"""Per-operator parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

import mistral_oracle as mo
from hip_util import bf16_ulp_close

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _hip():
    from mistral_inference import _hip
    return _hip


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("T,D", [(1, 4096), (3, 256), (40, 5120), (7, 14336)])
def test_rmsnorm(T, D):
    h = _hip()
    x, w = rnd(T, D, seed=1, scale=3.0), (1 + 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(2))).to(BF)
    ref = mo.rms_norm(x, w, 1e-5)
    got = h.rmsnorm(x.cuda(), w.cuda(), 1e-5).cpu()
    mism = (got.view(torch.int16) != ref.view(torch.int16)).float().mean().item()
    ok, err = bf16_ulp_close(got, ref, ulps=1.0)
    assert ok and mism < 2e-3, (mism, err)  # reduction order may flip a rounding on a handful of elements


def test_embedding():
    h = _hip()
    tab = rnd(1000, 512, seed=3)
    ids = torch.tensor([0, 999, 5, 5, 123], dtype=torch.long)
    assert torch.equal(h.embedding(tab.cuda(), ids.cuda()).cpu(), tab[ids])


def test_rope_bit_exact():
    h = _hip()
    T, H, Hkv, Dh = 37, 4, 2, 128
    qkv = rnd(T, (H + 2 * Hkv) * Dh, seed=4)
    cs = mo.rope_angles(Dh, 5000, 1e6)
    pos = torch.randint(0, 5000, (T,), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
    q = mo.apply_rope(qkv[:, : H * Dh].reshape(T, H, Dh), cs[pos.long()]).reshape(T, -1)
    k = mo.apply_rope(qkv[:, H * Dh: (H + Hkv) * Dh].reshape(T, Hkv, Dh), cs[pos.long()]).reshape(T, -1)
    dev = qkv.cuda()
    h.rope_inplace(dev, H, Hkv, Dh, cs.cuda(), pos.cuda())
    got = dev.cpu()
    assert torch.equal(got[:, : H * Dh], q)
    assert torch.equal(got[:, H * Dh: (H + Hkv) * Dh], k)
    assert torch.equal(got[:, (H + Hkv) * Dh:], qkv[:, (H + Hkv) * Dh:])  # v untouched


def _lin_ref(x, ws):
    return torch.cat([F.linear(x.float(), w.float()) for w in ws], dim=1)


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 9, 64, 200])
@pytest.mark.parametrize("K,N", [(256, 512), (4096, 1024), (1024, 4096), (14336, 512), (28672, 256)])
def test_linear_store(M, K, N):
    h = _hip()
    x, w = rnd(M, K, seed=6), rnd(N, K, seed=7, scale=1 / math.sqrt(K))
    ref = _lin_ref(x, [w])
    got = h.linear(x.cuda(), (w.cuda(),), h.EPI_STORE).cpu()
    ok, err = bf16_ulp_close(got, ref.to(BF), ulps=1.0)
    assert ok, err


@pytest.mark.parametrize("M", [1, 4, 150])
def test_linear_three_segments_and_transpose_detection(M):
    """q|k|v style row blocks, asymmetric weights (a transposed or permuted C write cannot pass)."""
    h = _hip()
    K = 512
    x = rnd(M, K, seed=8)
    ws = [rnd(512, K, seed=9, scale=0.05), rnd(256, K, seed=10, scale=0.05), rnd(256, K, seed=11, scale=0.05)]
    ws[0][3, :] += 0.5  # make rows distinguishable
    ref = _lin_ref(x, ws).to(BF)
    got = h.linear(x.cuda(), tuple(w.cuda() for w in ws), h.EPI_STORE).cpu()
    ok, err = bf16_ulp_close(got, ref, ulps=1.0)
    assert ok, err


@pytest.mark.parametrize("M", [1, 3, 8, 130])
def test_linear_residual_swiglu_logits(M):
    h = _hip()
    K, N = 512, 1024
    x = rnd(M, K, seed=12)
    w1, w3 = rnd(N, K, seed=13, scale=0.06), rnd(N, K, seed=14, scale=0.06)
    res = rnd(M, N, seed=15)
    # residual
    y = F.linear(x, w1)
    ref = (res + y).float()
    got = h.linear(x.cuda(), (w1.cuda(),), h.EPI_RESIDUAL, residual=res.cuda()).cpu()
    # bf16(W x) may be 1 ulp off (fp32 summation order); after the add that ulp is relative to |W x|, not to the sum
    scale = torch.maximum(torch.maximum(res.float().abs(), y.float().abs()), ref.abs())
    assert bool(((got.float() - ref).abs() <= 1.5 * scale * 2.0 ** -7 + 1e-6).all())
    # swiglu with the reference's rounding chain
    ref = (F.silu(F.linear(x, w1)) * F.linear(x, w3)).float()
    got = h.linear(x.cuda(), (w1.cuda(), w3.cuda()), h.EPI_SWIGLU).cpu()
    # a = bf16(W1 x) and b = bf16(W3 x) may each land 1 ulp away from the CPU's (fp32 summation order), and that
    # ulp is relative to |a|, |b| ~ 1, not to the (possibly small) product: relative + absolute tolerance
    g, r = got.float(), ref.float()
    err = (g - r).abs()
    assert bool((err <= 4 * r.abs() * 2.0 ** -7 + 8e-3).all()), float(err.max())
    # logits: fp32 tensor of bf16-rounded values
    got = h.linear(x.cuda(), (w1.cuda(),), h.EPI_LOGITS).cpu()
    assert got.dtype == torch.float32 and torch.equal(got, got.to(BF).float())
    assert bf16_ulp_close(got, F.linear(x, w1).float(), ulps=1.0)[0]


@pytest.mark.parametrize("M", [256, 300, 1030])
@pytest.mark.parametrize("K,N", [(128, 256), (512, 640), (4096, 1024)])
def test_linear_large_tile(M, K, N):
    """M >= 256 takes the 256x256 LDS-DMA kernel (gemm256.hip): every epilogue, ragged M and N edges, q|k|v style row
    segments, and a repeated-launch screen (the K loop keeps DMAs in flight across barriers; a hazard there would show
    up as run-to-run differences)."""
    h = _hip()
    x = rnd(M, K, seed=30)
    n0 = N // 2
    ws = [rnd(n0, K, seed=31, scale=1 / math.sqrt(K)), rnd(N - n0 - 64, K, seed=32, scale=1 / math.sqrt(K)),
          rnd(64, K, seed=33, scale=1 / math.sqrt(K))]
    ws[0][5, :] += 0.25
    xc, wc = x.cuda(), tuple(w.cuda() for w in ws)
    ref = _lin_ref(x, ws)
    got = h.linear(xc, wc, h.EPI_STORE)
    ok, err = bf16_ulp_close(got.cpu(), ref.to(BF), ulps=1.0)
    assert ok, err
    for _ in range(10):
        assert torch.equal(h.linear(xc, wc, h.EPI_STORE), got)
    # logits
    lg = h.linear(xc, wc, h.EPI_LOGITS).cpu()
    assert lg.dtype == torch.float32 and torch.equal(lg.to(BF), got.cpu())
    # residual
    res = rnd(M, N, seed=34)
    y = ref.to(BF)
    r2 = (res + y).float()
    g2 = h.linear(xc, wc, h.EPI_RESIDUAL, residual=res.cuda()).cpu().float()
    scale = torch.maximum(torch.maximum(res.float().abs(), y.float().abs()), r2.abs())
    assert bool(((g2 - r2).abs() <= 1.5 * scale * 2.0 ** -7 + 1e-6).all())
    # swiglu: W1 = first N/2 rows... use two equal-height matrices
    w1, w3 = rnd(N, K, seed=35, scale=1 / math.sqrt(K)), rnd(N, K, seed=36, scale=1 / math.sqrt(K))
    r3 = (F.silu(F.linear(x, w1)) * F.linear(x, w3)).float()
    g3 = h.linear(xc, (w1.cuda(), w3.cuda()), h.EPI_SWIGLU).cpu().float()
    e3 = (g3 - r3).abs()
    assert bool((e3 <= 4 * r3.abs() * 2.0 ** -7 + 8e-3).all()), float(e3.max())


def test_linear_tail_round_split():
    """16 x 24 = 384 tiles of 256x256 would run 1.5 rounds of the 256 CUs: launch_gemm gives the first 4096 columns to the
    256-tile kernel and the rest to the 128-tile kernel.  Row segments are chosen so that the cut falls INSIDE a segment
    and the tail spans two; every epilogue; SwiGLU splits at column 2048 of 3072."""
    h = _hip()
    M, K, N = 4096, 128, 6144
    x = rnd(M, K, seed=40)
    ws = [rnd(3000, K, seed=41, scale=0.09), rnd(1500, K, seed=42, scale=0.09), rnd(1644, K, seed=43, scale=0.09)]
    xc, wc = x.cuda(), tuple(w.cuda() for w in ws)
    ref = _lin_ref(x, ws)
    got = h.linear(xc, wc, h.EPI_STORE)
    ok, err = bf16_ulp_close(got.cpu(), ref.to(BF), ulps=1.0)
    assert ok, err
    lg = h.linear(xc, wc, h.EPI_LOGITS).cpu()
    assert torch.equal(lg.to(BF), got.cpu())
    res = rnd(M, N, seed=44)
    y = ref.to(BF)
    r2 = (res + y).float()
    g2 = h.linear(xc, wc, h.EPI_RESIDUAL, residual=res.cuda()).cpu().float()
    scale = torch.maximum(torch.maximum(res.float().abs(), y.float().abs()), r2.abs())
    assert bool(((g2 - r2).abs() <= 1.5 * scale * 2.0 ** -7 + 1e-6).all())
    F_ = 3072
    w1, w3 = rnd(F_, K, seed=45, scale=0.09), rnd(F_, K, seed=46, scale=0.09)
    r3 = (F.silu(F.linear(x, w1)) * F.linear(x, w3)).float()
    g3 = h.linear(xc, (w1.cuda(), w3.cuda()), h.EPI_SWIGLU).cpu().float()
    e3 = (g3 - r3).abs()
    assert bool((e3 <= 4 * r3.abs() * 2.0 ** -7 + 8e-3).all()), float(e3.max())


@pytest.mark.parametrize("M", [4096, 4000])
@pytest.mark.parametrize("K", [128, 192, 448, 4096])
def test_linear_half_height_tail_tiles(M, K):
    """The partly filled last round of a 256-tile launch runs as 128 x 256 tiles of the same kernel (three-stage ring, two
    phases per K tile: launch_gemm256_half).  16 x 24 = 384 tiles -> columns 4096.. are the tail.  Same MFMA shape and k
    order as the square tile, so the tail columns must equal, bit for bit, (a) the same columns computed as a problem of
    their own (128 square tiles: no split) and (c) the unsplit launch, and agree with (b) the tail on the 128-tile kernel; K from
    two K tiles (prologue only) over odd counts (ring wrap at every phase) to the real 64; ragged last m-tile (M = 4000);
    store and residual epilogues; repeated launches (DMAs stay in flight across barriers)."""
    h = _hip()
    N = 6144
    x = rnd(M, K, seed=50)
    ws = [rnd(4096, K, seed=51, scale=1 / math.sqrt(K)), rnd(1024, K, seed=52, scale=1 / math.sqrt(K)),
          rnd(1024, K, seed=53, scale=1 / math.sqrt(K))]
    res = rnd(M, N, seed=54)
    xc, wc, rc = x.cuda(), tuple(w.cuda() for w in ws), res.cuda()
    try:
        h.debug_set_prefill_kernels(gemm_tail=2)
        got = h.linear(xc, wc, h.EPI_STORE)
        for _ in range(5):
            assert torch.equal(h.linear(xc, wc, h.EPI_STORE), got)
        got_r = h.linear(xc, wc, h.EPI_RESIDUAL, residual=rc)
        alone = h.linear(xc, wc[1:], h.EPI_STORE)                      # (a) 16 x 8 tiles: one launch of the square kernel
        assert torch.equal(got[:, 4096:], alone)
        alone_r = h.linear(xc, wc[1:], h.EPI_RESIDUAL, residual=rc[:, 4096:].contiguous())
        assert torch.equal(got_r[:, 4096:], alone_r)
        h.debug_set_prefill_kernels(gemm_tail=0)                       # (c) unsplit: 1.5 rounds of the square kernel
        assert torch.equal(h.linear(xc, wc, h.EPI_STORE), got)
        assert torch.equal(h.linear(xc, wc, h.EPI_RESIDUAL, residual=rc), got_r)
        h.debug_set_prefill_kernels(gemm_tail=1)                       # (b) tail on the 128-tile kernel (A x W operand order)
        ok, err = bf16_ulp_close(h.linear(xc, wc, h.EPI_STORE).cpu(), got.cpu(), ulps=1.0)
        assert ok, err
    finally:
        h.debug_set_prefill_kernels(gemm_tail=2)
    if K <= 448:  # (the CPU matmul of the full K is minutes)
        ok, err = bf16_ulp_close(got.cpu(), _lin_ref(x, ws).to(BF), ulps=1.0)
        assert ok, err


@pytest.mark.parametrize("M", [3, 40, 300, 1030])
@pytest.mark.parametrize("K,V", [(256, 1000), (512, 4096), (200, 1000)])  # K = 200: never the fused route (K % 64)
def test_lm_head_logprobs(M, K, V):
    """log_softmax(logits)[m, target[m]] in one pass over the LM head (fused epilogue for M >= 256, logits + row kernel
    below) against torch's log_softmax of the SAME bf16-rounded logits; ragged vocab tail, ignored rows."""
    h = _hip()
    x, w = rnd(M, K, seed=50), rnd(V, K, seed=51, scale=4 / math.sqrt(K))
    tgt = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(52), dtype=torch.int32)
    tgt[0] = V - 1
    tgt[M // 2] = -1  # ignored row: returns -logsumexp
    got = h.lm_head_logprobs(x.cuda(), w.cuda(), tgt.cuda()).cpu()
    logits = h.linear(x.cuda(), (w.cuda(),), h.EPI_LOGITS).cpu()
    lsm = torch.log_softmax(logits, dim=-1)
    ref = torch.where(tgt >= 0, lsm.gather(1, tgt.clamp(min=0).long()[:, None])[:, 0], -torch.logsumexp(logits, dim=-1))
    assert float((got - ref).abs().max()) <= 2e-4, float((got - ref).abs().max())
    assert bf16_ulp_close(logits, F.linear(x, w).float(), ulps=1.0)[0]


@pytest.mark.parametrize("M", [1, 4])
def test_linear_fused_norm(M):
    h = _hip()
    K, N = 1024, 512
    x, w = rnd(M, K, seed=16, scale=2.0), rnd(N, K, seed=17, scale=0.03)
    nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(18))).to(BF)
    ref = F.linear(mo.rms_norm(x, nw, 1e-5), w).float()
    got = h.linear(x.cuda(), (w.cuda(),), h.EPI_STORE, norm_w=nw.cuda(), eps=1e-5).cpu()
    assert bf16_ulp_close(got, ref, ulps=1.5)[0]


def _ring_case(B, W, Hkv, Dh, lens, seed):
    """Fill rings as a model would: token p of sequence b sits at slot p % W."""
    g = torch.Generator().manual_seed(seed)
    ck = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    cv = torch.zeros(B, W, Hkv, Dh, dtype=BF)
    hist_k, hist_v = [], []
    for b, n in enumerate(lens):
        k = torch.randn(n, Hkv, Dh, generator=g).to(BF)
        v = torch.randn(n, Hkv, Dh, generator=g).to(BF)
        for p in range(n):
            ck[b, p % W], cv[b, p % W] = k[p], v[p]
        hist_k.append(k)
        hist_v.append(v)
    return ck, cv, hist_k, hist_v


@pytest.mark.parametrize("H,Hkv", [(4, 2), (32, 8), (8, 8), (12, 2), (16, 2), (24, 2), (6, 2), (10, 2), (32, 2)])
@pytest.mark.parametrize("W,lens", [(16, [5, 16, 40]), (300, [1, 299, 300]), (4096, [4096, 17, 5000]), (9000, [9000, 5000])])
def test_attn_decode(H, Hkv, W, lens):
    """lens[b] = tokens seen INCLUDING the new one (already in the ring)."""
    h = _hip()
    Dh, B = 128, len(lens)
    ck, cv, hk, hv = _ring_case(B, W, Hkv, Dh, lens, seed=20)
    q = rnd(B, H * Dh, seed=21)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32)
    got = h.attn_decode(q.cuda(), ck.cuda(), cv.cuda(), H, pos.cuda()).cpu()
    for b, n in enumerate(lens):
        lo = max(0, n - W)
        kp = torch.arange(lo, n)
        ref = mo._attend(q[b].view(1, H, Dh), hk[b][lo:n], hv[b][lo:n], torch.tensor([n - 1]), kp, W, causal=True)
        ok, err = bf16_ulp_close(got[b], ref[0], ulps=2.0, floor=1e-2)
        assert ok, (b, err)
    # the arrival counters must be left at zero: a second call gives the same answer
    again = h.attn_decode(q.cuda(), ck.cuda(), cv.cuda(), H, pos.cuda()).cpu()
    assert torch.equal(got, again)


@pytest.mark.parametrize("H,Hkv", [(4, 2), (4, 4), (32, 8), (12, 2)])
@pytest.mark.parametrize("W", [5000, 8192, 16384])
def test_attn_decode_wide_ring_constant_v(H, Hkv, W):
    """V == 1 everywhere: softmax weights sum to one, so every output element must be exactly 1.0 whatever K is, and
    two launches must agree bit for bit.  A size-independent property for rings the oracle is too slow for; it is the
    case that exposed in-flight register corruption in an earlier hand-scheduled load variant."""
    h = _hip()
    g = torch.Generator().manual_seed(W + H)
    ck = torch.randn(1, W, Hkv, 128, generator=g).to(BF).cuda()
    cv = torch.ones(1, W, Hkv, 128, dtype=BF).cuda()
    q = torch.randn(1, H * 128, generator=g).to(BF).cuda()
    for n in (W, W // 2 + 3, 3 * W):
        pos = torch.tensor([n - 1], dtype=torch.int32).cuda()
        a = h.attn_decode(q, ck, cv, H, pos)
        b = h.attn_decode(q, ck, cv, H, pos)
        assert torch.equal(a, b)
        assert bool((a.float() == 1.0).all()), (n, float((a.float() - 1).abs().max()))


@pytest.mark.parametrize("H,Hkv", [(4, 2), (32, 8)])
@pytest.mark.parametrize("W,seen,new", [
    (4096, [0, 0, 0], [7, 150, 33]),      # first prefill, ragged
    (8, [0, 0], [13, 5]),                 # window shorter than the prompt
    (8, [4, 8, 21], [4, 4, 2]),           # later chunk: ring + new keys, wrap
    (64, [100], [200]),                   # chunk longer than the window over a wrapped ring
    (512, [0], [700]),                    # several key tiles, window cuts early tiles
    (4096, [0, 0], [1100, 900]),          # with 32 heads: enough 256-query blocks for the 8-wave kernel, ragged tail
    (512, [600, 30, 0], [700, 520, 300]), # 8-wave kernel over wrapped rings with the window cutting tiles
    (4096, [0], [4096]),                  # THE headline prefill shape (with 32/8 heads): one 4096-token sequence, W = 4096
    (4096, [4096], [2048]),               # second chunk over a full ring: every query sees exactly 4096 keys
])
def test_attn_prefill(H, Hkv, W, seen, new):
    h = _hip()
    Dh, B = 128, len(new)
    T = sum(new)
    ck, cv, hk, hv = _ring_case(B, W, Hkv, Dh, seen, seed=22)
    qkv = rnd(T, (H + 2 * Hkv) * Dh, seed=23)
    q_start = torch.tensor([0] + list(torch.tensor(new).cumsum(0)), dtype=torch.int32)
    kv_before = torch.tensor(seen, dtype=torch.int32)
    got = h.attn_prefill(qkv.cuda(), H, Hkv, Dh, ck.cuda(), cv.cuda(), W, q_start.cuda(), kv_before.cuda(), B, max(new)).cpu()
    nq, nkv = H * Dh, Hkv * Dh
    o = 0
    for b, s in enumerate(new):
        p = seen[b]
        rows = qkv[o:o + s]
        n_old = min(p, W)
        keys = torch.cat([hk[b][p - n_old:p], rows[:, nq:nq + nkv].reshape(s, Hkv, Dh)])
        vals = torch.cat([hv[b][p - n_old:p], rows[:, nq + nkv:].reshape(s, Hkv, Dh)])
        kpos = torch.arange(p - n_old, p + s)
        if s * keys.shape[0] * H > 2 ** 28:   # [H, s, n] fp32 scores would be GBs: one kv head (and its q heads) at a time
            R = H // Hkv
            qh = rows[:, :nq].reshape(s, Hkv, R, Dh)
            ref = torch.cat([mo._attend(qh[:, g], keys[:, g:g + 1], vals[:, g:g + 1], torch.arange(p, p + s), kpos, W, causal=True)
                             .view(s, R, Dh) for g in range(Hkv)], dim=1).reshape(s, nq)
        else:
            ref = mo._attend(rows[:, :nq].reshape(s, H, Dh), keys, vals, torch.arange(p, p + s), kpos, W, causal=True)
        # P is rounded to bf16 for the P.V MFMA (as in every flash kernel, xformers' included; SURVEY.md
        # Appendix A): absolute error <= ~2^-9 * max|V| + one output rounding, independent of |out|
        err = (got[o:o + s].float() - ref.float()).abs().max().item()
        assert err <= 2.5e-2, (b, err)
        o += s


def test_attn_prefill_nocache_unmasked():
    h = _hip()
    H, Hkv, Dh, T = 4, 2, 128, 150
    qkv = rnd(T, (H + 2 * Hkv) * Dh, seed=24)
    got = h.attn_prefill(qkv.cuda(), H, Hkv, Dh, None, None, T, None, None, 1, T, causal=False).cpu()
    nq, nkv = H * Dh, Hkv * Dh
    pos = torch.arange(T)
    ref = mo._attend(qkv[:, :nq].reshape(T, H, Dh), qkv[:, nq:nq + nkv].reshape(T, Hkv, Dh),
                     qkv[:, nq + nkv:].reshape(T, Hkv, Dh), pos, pos, None, causal=False)
    err = (got.float() - ref.float()).abs().max().item()
    assert err <= 2.5e-2, err


def test_kv_write_keeps_last_window():
    h = _hip()
    B, W, Hkv, Dh = 2, 4, 2, 128
    new, seen = [6, 3], [5, 2]
    T = sum(new)
    k, v = rnd(T, Hkv * Dh, seed=25), rnd(T, Hkv * Dh, seed=26)
    ck = torch.full((B, W, Hkv, Dh), 7.0, dtype=BF)
    cv = torch.full((B, W, Hkv, Dh), 9.0, dtype=BF)
    tok_seq = torch.tensor([0] * 6 + [1] * 3, dtype=torch.int32)
    tok_pos = torch.tensor([5 + i for i in range(6)] + [2 + i for i in range(3)], dtype=torch.int32)
    q_start = torch.tensor([0, 6, 9], dtype=torch.int32)
    dk, dv = ck.cuda(), cv.cuda()
    h.kv_write(dk, dv, k.cuda(), v.cuda(), tok_seq.cuda(), tok_pos.cuda(), q_start.cuda())
    ek, ev = ck.clone(), cv.clone()
    for t in range(T):
        b = int(tok_seq[t])
        i = t - int(q_start[b])
        if i >= new[b] - W:
            ek[b, int(tok_pos[t]) % W] = k[t].view(Hkv, Dh)
            ev[b, int(tok_pos[t]) % W] = v[t].view(Hkv, Dh)
    assert torch.equal(dk.cpu(), ek) and torch.equal(dv.cpu(), ev)


@pytest.mark.parametrize("T", [1, 5, 40])
def test_moe_router(T):
    h = _hip()
    D, E, k = 512, 8, 2
    x, gate = rnd(T, D, seed=27), rnd(E, D, seed=28, scale=0.05)
    logits = F.linear(x, gate)
    tw, ti = torch.topk(logits, k)
    tw = torch.softmax(tw, dim=1, dtype=torch.float).to(BF).float()
    idx, w = h.moe_router(x.cuda(), gate.cuda(), k)
    idx, w = idx.cpu().long(), w.cpu()
    for t in range(T):
        srt = torch.sort(logits[t].float(), descending=True).values
        if srt[k - 1] == srt[k] or (k > 1 and srt[0] == srt[1]):
            continue  # tie on bf16 logits: torch.topk's order is unspecified (hard parts)
        got_logits = logits[t][idx[t]].float()
        ref_logits = logits[t][ti[t]].float()
        if not torch.equal(idx[t], ti[t]):
            # accumulation-order noise may flip a bf16 rounding of a logit; require the same VALUES picked
            assert torch.allclose(got_logits, ref_logits, atol=2e-2), (t, idx[t], ti[t])
        assert torch.allclose(w[t], tw[t], atol=8e-3), (t, w[t], tw[t])


# ------------------------------------------------------------------------------------------------
# leaf entry points of the fused decode operators (SURVEY.md section 8b minimum export set)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T", [1, 3, 8])
@pytest.mark.parametrize("H,Hkv,D", [(4, 2, 512), (32, 8, 4096), (8, 8, 1024), (8, 2, 2560), (32, 8, 5120)])  # (the last two: rows of 5 / 10 chunks = the 10-load batches at T = 1)
def test_qkv_rope_kvwrite_leaf(T, H, Hkv, D):
    """mi_qkv_rope_kvwrite = RMSNorm + q|k|v GEMV + RoPE + ring write vs the oracle's pieces (transformer_layers.py:66-70,
    rope.py:13-23, cache.py:83-92).  Projections are compared to 1 bf16 ulp of the pre-rotation value (summation order),
    the ring rows must be exactly the k/v columns of the returned activations."""
    h = _hip()
    Dh, W, B = 128, 24, T
    x = rnd(T, D, seed=40, scale=2.0)
    nw = (1 + 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(41))).to(BF)
    wq, wk, wv = (rnd(n, D, seed=42 + i, scale=D ** -0.5) for i, n in enumerate((H * Dh, Hkv * Dh, Hkv * Dh)))
    cs = mo.rope_angles(Dh, 4000, 1e6)
    pos = torch.randint(0, 4000, (T,), generator=torch.Generator().manual_seed(45), dtype=torch.int32)
    seq = torch.arange(T, dtype=torch.int32).flip(0).contiguous()  # rows deliberately not in token order
    ck = torch.zeros(B, W, Hkv, Dh, dtype=BF).cuda()
    cv = torch.zeros_like(ck)
    got = h.qkv_rope_kvwrite(x.cuda(), wq.cuda(), wk.cuda(), wv.cuda(), Dh, cs.cuda(), pos.cuda(), norm_w=nw.cuda(), eps=1e-5,
                             cache_k=ck, cache_v=cv, tok_seq=seq.cuda()).cpu()
    xn = mo.rms_norm(x, nw, 1e-5)
    q = mo.apply_rope(F.linear(xn, wq).reshape(T, H, Dh), cs[pos.long()]).reshape(T, -1)
    k = mo.apply_rope(F.linear(xn, wk).reshape(T, Hkv, Dh), cs[pos.long()]).reshape(T, -1)
    v = F.linear(xn, wv)
    ref = torch.cat([q, k, v], dim=1)
    # a rotated pair mixes two projections: bound the error by 1.5 ulp of the larger member of the pair
    mag = ref.float().abs().reshape(T, -1, 2).amax(-1, keepdim=True).expand(-1, -1, 2).reshape(T, -1)
    err = (got.float() - ref.float()).abs()
    assert bool((err <= 1.5 * torch.clamp(mag, min=1e-2) * 2.0 ** -7 + 1e-6).all()), float(err.max())
    assert float((err > 0).float().mean()) < 0.2
    for t in range(T):
        slot = int(pos[t]) % W
        assert torch.equal(ck[int(seq[t]), slot].cpu().reshape(-1), got[t, H * Dh:(H + Hkv) * Dh])
        assert torch.equal(cv[int(seq[t]), slot].cpu().reshape(-1), got[t, (H + Hkv) * Dh:])
    # without a cache and without the fused norm the same call is projection + RoPE only
    got2 = h.qkv_rope_kvwrite(xn.cuda(), wq.cuda(), wk.cuda(), wv.cuda(), Dh, cs.cuda(), pos.cuda()).cpu()
    assert float((got2.float() - ref.float()).abs().max()) <= float(err.max()) + 4e-2


def _moe_case(T, D, Fh, E, k, seed):
    x = rnd(T, D, seed=seed, scale=1.0)
    gate = rnd(E, D, seed=seed + 1, scale=0.05)
    experts = [(rnd(Fh, D, seed=seed + 10 + 3 * e, scale=D ** -0.5), rnd(D, Fh, seed=seed + 11 + 3 * e, scale=Fh ** -0.5),
                rnd(Fh, D, seed=seed + 12 + 3 * e, scale=D ** -0.5)) for e in range(E)]
    return x, gate, experts


@pytest.mark.parametrize("T", [1, 2, 8, 9, 300])
def test_moe_expert_leaves_vs_oracle(T):
    """mi_moe_experts_decode (T <= 8) and mi_moe_grouped_gemm (T > 8) behind `_hip.moe_experts`, with and without a
    residual, against oracle moe_ffn (moe.py:24-32).  Tokens whose k-th/(k+1)-th router logits tie within 2 ulp are
    excluded (torch.topk's tie order is unspecified)."""
    h = _hip()
    D, Fh, E, k = 512, 1024, 8, 2
    x, gate, experts = _moe_case(T, D, Fh, E, k, seed=60)
    mo.ROUTER_TRACE = []
    ref = mo.moe_ffn(x, gate, experts, k)
    logits, mo.ROUTER_TRACE = mo.ROUTER_TRACE[0], None
    srt = torch.sort(logits, dim=1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 2 * srt[:, k - 1].abs().clamp(min=1e-3) * 2.0 ** -7
    assert int(clear.sum()) >= max(1, int(0.7 * T))
    dev = [tuple(w.cuda() for w in ex) for ex in experts]
    tab = torch.tensor([[w.data_ptr() for w in ex] for ex in dev], dtype=torch.int64, device="cuda")
    idx, w = h.moe_router(x.cuda(), gate.cuda(), k)
    got = h.moe_experts(x.cuda(), tab, E, Fh, idx, w).cpu()
    tol = 3e-2 * max(1.0, float(ref.float().abs().max()))
    assert float((got.float() - ref.float())[clear].abs().max()) <= tol
    res = rnd(T, D, seed=99, scale=2.0)
    got_r = h.moe_experts(x.cuda(), tab, E, Fh, idx, w, residual=res.cuda()).cpu()
    assert torch.equal(got_r[clear], (res + got)[clear])  # bf16(h + R): the block's residual add (transformer_layers.py:168)


def test_moe_layer_module_uses_the_native_kernels():
    """MoeLayer.forward holds no torch compute: its result equals the leaf call on the same router output."""
    h = _hip()
    from torch import nn
    from mistral_inference.args import MoeArgs
    from mistral_inference.moe import MoeLayer
    from mistral_inference.transformer_layers import FeedForward
    D, Fh, E, k = 512, 1024, 4, 2
    x, gate, experts = _moe_case(5, D, Fh, E, k, seed=80)
    layer = MoeLayer([FeedForward(D, Fh) for _ in range(E)], nn.Linear(D, E, bias=False), MoeArgs(num_experts=E, num_experts_per_tok=k))
    layer = layer.to(BF).cuda()
    with torch.no_grad():
        layer.gate.weight.copy_(gate)
        for ex, (w1, w2, w3) in zip(layer.experts, experts):
            ex.w1.weight.copy_(w1)
            ex.w2.weight.copy_(w2)
            ex.w3.weight.copy_(w3)
        out = layer(x.cuda()).cpu()
    ref = mo.moe_ffn(x, gate, experts, k)
    assert float((out.float() - ref.float()).abs().max()) <= 3e-2 * max(1.0, float(ref.float().abs().max()))


@pytest.mark.parametrize("E,k,T", [(6, 3, 5), (20, 2, 33), (4, 4, 3)])
def test_moe_layer_general_route_any_expert_count_and_topk(E, k, T):
    """The reference accepts any num_experts / num_experts_per_tok (moe.py:24-32); the fused kernels take E <= 16 and
    k in {1, 2, 4}.  Everything else runs the reference's loop with the dense work on mi_linear (MoeLayer._forward_general)
    instead of raising.  (4, 4, 3) is a fused shape: both routes must agree on it.)"""
    from torch import nn
    from mistral_inference.args import MoeArgs
    from mistral_inference.moe import MoeLayer
    from mistral_inference.transformer_layers import FeedForward
    D, Fh = 512, 1024
    x, gate, experts = _moe_case(T, D, Fh, E, k, seed=90 + E)
    layer = MoeLayer([FeedForward(D, Fh) for _ in range(E)], nn.Linear(D, E, bias=False), MoeArgs(num_experts=E, num_experts_per_tok=k))
    layer = layer.to(BF).cuda()
    with torch.no_grad():
        layer.gate.weight.copy_(gate)
        for ex, (w1, w2, w3) in zip(layer.experts, experts):
            ex.w1.weight.copy_(w1)
            ex.w2.weight.copy_(w2)
            ex.w3.weight.copy_(w3)
        out = layer(x.cuda()).cpu()
        gen = layer._forward_general(x.cuda()).cpu()
    mo.ROUTER_TRACE = []
    ref = mo.moe_ffn(x, gate, experts, k)
    lg, mo.ROUTER_TRACE = mo.ROUTER_TRACE[0], None
    srt = torch.sort(lg, dim=1, descending=True).values
    clear = torch.ones(T, dtype=torch.bool) if k >= E else (srt[:, k - 1] - srt[:, k]) > 2 * srt[:, k - 1].abs().clamp(min=1e-3) * 2.0 ** -7
    tol = 3e-2 * max(1.0, float(ref.float().abs().max()))
    assert clear.any()
    assert float((out.float() - ref.float())[clear].abs().max()) <= tol
    assert float((gen.float() - ref.float())[clear].abs().max()) <= tol


@pytest.mark.gpu
def test_fused_rope_epilogue_bit_equal_to_separate_pass(tmp_path):
    """RoPE as the q|k|v GEMM's epilogue (csrc/gemm256.hip, csrc/gemm.hip; transformer_layers.py:66-70) == RoPE as the separate
    pass it replaced (MI_FUSE_ROPE=0), bit for bit: prefill logits of every chunk and the K/V rings, on the 128-tile kernel
    alone (ragged 3-sequence batch), on the 256-tile kernel (Mistral-7B dims, 1000 + 300 tokens) and on its half-height tail
    tiles (4096 tokens: the k | v columns)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    dumps = []
    for fuse in ("1", "0"):
        out = tmp_path / f"fuse{fuse}.pt"
        env = dict(os.environ, MI_FUSE_ROPE=fuse)
        r = subprocess.run([sys.executable, os.path.join(here, "fused_rope_util.py"), str(out)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        dumps.append(torch.load(out))
    a, b = dumps
    assert a.keys() == b.keys() and len(a) > 10
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
