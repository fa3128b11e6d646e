"""BASELINE.json configs[1] at FULL size (Mistral-7B-v0.3 dims, 32 layers, 4096-token prompt, sliding_window=4096) - the
workload bench.py times.  The CPU oracle needs minutes per token here, so parity is checked through size-independent
properties of the path (the reference's own self-consistency tests, tests/test_generate.py:36-69 and :199-230, are of
this kind): one-shot prefill == chunked prefill == token-by-token decode on the same tokens, across the ring wrap at
position 4096; replayed-graph decode == launch-by-launch decode; and run-to-run bit equality."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16
TOL = 6e-2  # max-abs on logits between two schedules of the SAME model: 32 layers of bf16 storage (|logits| ~ 3)


@pytest.fixture(scope="module")
def full_model():
    sys.path.insert(0, ROOT)
    import bench
    model = bench.build_model(dict(bench.MISTRAL_7B), 0, 1, "cuda")
    yield model
    del model
    torch.cuda.empty_cache()


def _cache(model, n):
    from mistral_inference.cache import BufferCache
    a = model.args
    c = BufferCache(model.n_local_layers, 1, n, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=BF)
    c.reset()
    return c


def test_full_size_prefill_chunked_decode_agree(full_model):
    m = full_model
    T, extra = 4096, 6
    ids = torch.randint(0, m.args.vocab_size, (T + extra,), generator=torch.Generator().manual_seed(0)).cuda()
    # (a) one-shot prefill of 4096, then teacher-forced decode across the ring wrap (positions 4096..4101)
    c = _cache(m, T + extra)
    one = m.forward(ids[:T], [T], c)[-1].clone()
    dec = [m.forward(ids[T + i:T + i + 1], [1], c)[0].clone() for i in range(extra)]
    assert torch.isfinite(one).all() and all(torch.isfinite(d).all() for d in dec)
    # (b) chunked prefill 3 x ~1365 must give the same last-row logits
    c2 = _cache(m, T + extra)
    for lo in range(0, T, 1366):
        last = m.forward(ids[lo:min(lo + 1366, T)], [min(lo + 1366, T) - lo], c2)[-1]
    assert float((last - one).abs().max()) < TOL
    assert int(last.argmax()) == int(one.argmax()) or float(one.max() - one[last.argmax()]) < TOL
    # (c) the next tokens fed as ONE more chunk (prefill branch over a wrapped ring) == the decode steps
    chunk = m.forward(ids[T:T + extra], [extra], c2)
    for i in range(extra):
        assert float((chunk[i] - dec[i]).abs().max()) < TOL, i
    # (d) bit-exact repeatability of the whole schedule
    c3 = _cache(m, T + extra)
    again = m.forward(ids[:T], [T], c3)[-1]
    assert torch.equal(again, one)
    dec2 = [m.forward(ids[T + i:T + i + 1], [1], c3)[0].clone() for i in range(extra)]
    assert all(torch.equal(a, b) for a, b in zip(dec, dec2))


def test_full_size_graph_replay_equals_eager(full_model):
    m = full_model
    T, steps = 4096, 5
    ids = torch.randint(0, m.args.vocab_size, (T + steps,), generator=torch.Generator().manual_seed(1)).cuda()
    c = _cache(m, T + steps)
    m.forward(ids[:T], [T], c)
    eager = [m.forward(ids[T + i:T + i + 1], [1], c)[0].clone() for i in range(steps)]
    c2 = _cache(m, T + steps)
    m.forward(ids[:T], [T], c2)
    with m.graphed_decode(c2):
        graph = [m.forward(ids[T + i:T + i + 1], [1], c2)[0].clone() for i in range(steps)]
    assert all(torch.equal(a, b) for a, b in zip(eager, graph))
    assert torch.equal(c.kv_seqlens, c2.kv_seqlens)


def test_full_size_prompt_logprobs_equal_forward(full_model):
    """The fused LM-head log-probability path generate() uses (no [T, V] logits) against forward() + log_softmax at
    T = 4096, V = 32768."""
    m = full_model
    T = 4096
    ids = torch.randint(0, m.args.vocab_size, (T,), generator=torch.Generator().manual_seed(2)).cuda()
    tgt = torch.cat([ids[1:], torch.tensor([-1], device="cuda")]).to(torch.int32)
    c = _cache(m, T + 2)
    logits = m.forward(ids, [T], c)
    ref = torch.log_softmax(logits, dim=-1)[torch.arange(T - 1, device="cuda"), ids[1:]]
    last_ref = logits[-1].clone()
    del logits
    c2 = _cache(m, T + 2)
    lp, last = m.prompt_logprobs(ids, [T], c2, tgt)
    assert float((lp[:-1] - ref).abs().max()) <= 1e-3
    # the last row goes through the GEMV kernels instead of the GEMM: same bf16 values up to one ulp of summation order
    d = (last[0] - last_ref).abs()
    assert float((d / last_ref.abs().clamp(min=1e-2)).max()) <= 2.0 ** -7 and float((d > 0).float().mean()) < 0.2


def test_full_size_greedy_soak_engine_equals_launch_path(full_model):
    """300 greedy steps behind a 4096-token prompt (the 4096-slot rings wrap from the first step on), the generate() loop body
    (reference generate.py:120-140 at temperature 0) once on the persistent engine and once on the launch path: every token and
    the last logits row identical, log-probabilities equal up to the log-sum-exp's fp32 summation order, status words clean.
    (scripts/soak_fullsize.py runs the same for 2000 steps and for the other BASELINE configs: profiles/r06b_soak_fullsize_*.)"""
    from mistral_inference import _hip
    m = full_model
    T, steps = 4096, 300
    prompt = torch.randint(0, m.args.vocab_size, (T,), generator=torch.Generator().manual_seed(5)).cuda()
    out = {}
    for engine in (True, False):
        prev = _hip.set_decode_engine(engine)
        try:
            c = _cache(m, T + steps + 8)
            last = m.forward(prompt, [T], c)[-1:]
            sess = m.greedy_session(c, torch.argmax(last, dim=-1))
            sess.run(steps)
            toks, lps = sess.collect()
            st = _hip.decode_engine_status(m._backend._workspace)
            assert st["status"] == 0 and st["abort"] == 0 and st["bad_id"] == 0, st
            out[engine] = (toks.cpu(), lps.cpu(), sess.logits.clone().cpu(), st["engine_launches"])
        finally:
            _hip.set_decode_engine(prev)
    assert out[True][3] >= steps and out[False][3] == out[True][3]  # (the engine ran the first loop; the second launched none)
    assert torch.equal(out[True][0], out[False][0]), int((out[True][0] != out[False][0]).any(dim=1).nonzero()[0, 0])
    assert torch.equal(out[True][2], out[False][2])
    assert float((out[True][1] - out[False][1]).abs().max()) < 2e-5
    assert out[True][0].unique().numel() > 50  # (a real continuation, not a fixed point)
