"""Greedy sampling fused into the decode step (mi_batch_t ABI v4, Transformer.greedy_session) against the loop it replaces:
`next = torch.argmax(logits); lp = torch.log_softmax(logits)[next]; logits = model.forward(next)` (reference
generate.py:124-140 at temperature 0).  Tokens must be IDENTICAL (same logits, same first-maximum tie rule), logprobs equal
up to fp32 summation order; on the persistent engine, on the launch path (MoE, batch > 1), eager and from the hipGraph; plus
the engine's residency gate (a launch that cannot have all workgroups resident writes nothing and the steps are re-run on
the launch path)."""
import pytest
import torch

import mistral_oracle as mo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

DENSE = dict(dim=512, n_layers=3, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
             vocab_size=1000, sliding_window=48)
MOE = dict(dim=512, n_layers=2, head_dim=128, hidden_dim=1024, n_heads=8, n_kv_heads=2, norm_eps=1e-5,
           vocab_size=777, sliding_window=None, moe=dict(num_experts=4, num_experts_per_tok=2))


def _model(p, seed, max_batch=1):
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    args = mo.OracleArgs.from_params(p)
    w = mo.synth_weights(args, seed=seed)
    targs = TransformerArgs.from_dict(mo.params_json(args))
    targs.max_batch_size = max_batch
    with torch.device("meta"):
        m = Transformer(targs)
    m = m.to(BF).to_empty(device="cuda")
    m.load_state_dict({k: v.cuda() for k, v in w.items()}, assign=True)
    return m.eval()


def _cache(m, B, n):
    from mistral_inference.cache import BufferCache
    a = m.args
    c = BufferCache(m.n_local_layers, a.max_batch_size, n, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=BF)
    c.reset()
    return c


def _prefill(m, prompts, total):
    c = _cache(m, len(prompts), total)
    flat = torch.tensor(sum(prompts, []), device="cuda")
    logits = m.forward(flat, [len(p) for p in prompts], c)
    ends = torch.tensor([len(p) for p in prompts], device="cuda").cumsum(0) - 1
    return c, logits.index_select(0, ends)


def _loop_reference(m, prompts, steps):
    """The unfused loop: forward() + torch.argmax + torch.log_softmax per token."""
    c, last = _prefill(m, prompts, max(len(p) for p in prompts) + steps + 2)
    toks, lps = [], []
    for _ in range(steps):
        nxt = torch.argmax(last, dim=-1)
        lps.append(torch.log_softmax(last, dim=-1).gather(1, nxt[:, None])[:, 0].clone())
        toks.append(nxt.clone())
        last = m.forward(nxt, [1] * len(prompts), c)
    return torch.stack(toks), torch.stack(lps)


def _session(m, prompts, steps, graph):
    c, last = _prefill(m, prompts, max(len(p) for p in prompts) + steps + 2)
    first = torch.argmax(last, dim=-1)
    lp0 = torch.log_softmax(last, dim=-1).gather(1, first[:, None])[:, 0]
    sess = m.greedy_session(c, first, graph=graph)
    sess.run(steps - 1)
    toks, lps = sess.collect()
    return torch.cat([first[None], toks]), torch.cat([lp0[None], lps]), sess


def _prompts(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, V, (n,), generator=g).tolist() for n in [37, 5, 18][:B]]


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("engine", [True, False])
def test_session_equals_forward_argmax_loop_dense(engine, graph):
    from mistral_inference import _hip
    m = _model(DENSE, seed=3)
    prev = _hip.set_decode_engine(engine)
    try:
        prompts = _prompts(1, DENSE["vocab_size"], 1)
        ref_t, ref_lp = _loop_reference(m, prompts, 20)   # crosses the 48-slot ring
        got_t, got_lp, sess = _session(m, prompts, 20, graph)
        st = _hip.decode_engine_status(m._backend._workspace)
        assert st["status"] == 0
        assert torch.equal(ref_t, got_t), (ref_t[:, 0].tolist(), got_t[:, 0].tolist())
        assert float((ref_lp - got_lp).abs().max()) < 2e-5
        # the logits of the last step are still produced (the LM head's work is unchanged)
        assert torch.isfinite(sess.logits).all() and int(torch.argmax(sess.logits, -1)[0]) >= 0
    finally:
        _hip.set_decode_engine(prev)


@pytest.mark.parametrize("graph", [False, True])
def test_session_moe_and_batch3_launch_path(graph):
    m = _model(MOE, seed=4, max_batch=3)
    prompts = _prompts(3, MOE["vocab_size"], 2)
    ref_t, ref_lp = _loop_reference(m, prompts, 9)
    got_t, got_lp, _ = _session(m, prompts, 9, graph)
    assert torch.equal(ref_t, got_t)
    assert float((ref_lp - got_lp).abs().max()) < 2e-5


def test_greedy_sample_leaf_ties_and_logprob():
    """First maximal index wins (torch.argmax), across threads, waves and the whole row; logprob = log_softmax at it."""
    from mistral_inference import _hip
    g = torch.Generator().manual_seed(0)
    V = 32768 + 5
    x = torch.randn(4, V, generator=g)
    x[0, [17, 4000, 30000]] = 9.0          # ties far apart: the lowest index
    x[1, [1023, 1024, 2047]] = 7.5         # ties in neighbouring threads / the same thread's next element
    x[2, V - 1] = 11.0                     # maximum in the ragged tail
    x[3] = torch.full((V,), -3.25)         # everything ties: index 0
    tok, lp = _hip.greedy_sample(x.cuda())
    assert tok.tolist() == [17, 1023, V - 1, 0]
    ref = torch.log_softmax(x, -1).gather(1, tok.cpu()[:, None])[:, 0]
    assert float((lp.cpu() - ref).abs().max()) < 2e-5
    assert tok.tolist() == torch.argmax(x, -1).tolist()


def test_engine_argmax_ties_lowest_index():
    """Duplicate LM-head rows make exact ties inside the engine's epilogue: rows in different waves of one workgroup and in
    different workgroups.  The sample must be the lowest index - what torch.argmax returns for the same logits."""
    from mistral_inference import _hip
    m = _model(DENSE, seed=5)
    V = DENSE["vocab_size"]
    with torch.no_grad():
        w = m.output.weight
        w[3] = w[3] * 4.0            # make one direction dominate so that the tie is at the maximum (or its negative:
        for j in (2, 410, 777, 998, 999):   # ... then the maximum is elsewhere and the test is vacuous for that step)
            w[j] = w[3]
    m._weights_changed()
    prompts = _prompts(1, V, 7)
    ref_t, _ = _loop_reference(m, prompts, 16)
    got_t, _, _ = _session(m, prompts, 16, graph=True)
    st = _hip.decode_engine_status(m._backend._workspace)
    assert st["engine_launches"] > 0
    assert torch.equal(ref_t, got_t)
    assert int((got_t == 2).sum()) >= 3   # the tied group did win several times, and its lowest member was reported


def test_generate_fused_equals_unfused_with_and_without_eos():
    from mistral_inference.generate import generate
    m = _model(DENSE, seed=6, max_batch=3)
    prompts = _prompts(3, DENSE["vocab_size"], 3)
    m.fused_greedy = False
    t0, l0 = generate(prompts, m, max_tokens=40, temperature=0.0)
    m.fused_greedy = True
    t1, l1 = generate(prompts, m, max_tokens=40, temperature=0.0)
    assert t0 == t1
    assert all(abs(a - b) < 1e-4 for x, y in zip(l0, l1) for a, b in zip(x, y))
    # EOS: every sequence must have produced it before the batch stops (generate.py:128-132); pick ids that occur
    for eos in {t0[0][3], t0[1][20], t0[2][35]}:
        m.fused_greedy = False
        a = generate(prompts, m, max_tokens=40, temperature=0.0, eos_id=eos)
        m.fused_greedy = True
        b = generate(prompts, m, max_tokens=40, temperature=0.0, eos_id=eos)
        assert a[0] == b[0], eos
        assert all(abs(x - y) < 1e-4 for r, s in zip(a[1], b[1]) for x, y in zip(r, s))
    # batch 1 on the engine, eos in the middle of a 32-step chunk and max_tokens = 1 / 2 edge cases
    one = [prompts[0]]
    for mt in (1, 2, 33, 70):
        m.fused_greedy = False
        a = generate(one, m, max_tokens=mt, temperature=0.0)
        m.fused_greedy = True
        b = generate(one, m, max_tokens=mt, temperature=0.0)
        assert a[0] == b[0], mt
    m.fused_greedy = False
    full = generate(one, m, max_tokens=70, temperature=0.0)[0][0]
    for eos in {full[0], full[1], full[40]}:
        m.fused_greedy = False
        a = generate(one, m, max_tokens=70, temperature=0.0, eos_id=eos)
        m.fused_greedy = True
        b = generate(one, m, max_tokens=70, temperature=0.0, eos_id=eos)
        assert a[0] == b[0], eos


def test_residency_gate_failure_on_the_first_sampled_step_keeps_the_input_token():
    """temperature > 0: the nucleus draw is its own small kernel behind the engine launch.  When the FIRST step of a session
    fails its residency gate the logits buffer has never been written: the draw must not run (it would replace the input id
    the recovery re-runs from by a sample of garbage).  Same seed, same tokens as an undisturbed session."""
    from mistral_inference import _hip
    m = _model(DENSE, seed=8)
    prompts = _prompts(1, DENSE["vocab_size"], 4)
    kw = dict(graph=True, temperature=0.7, top_p=0.8, seed=1234)
    try:
        c, last = _prefill(m, prompts, 60)
        first = torch.argmax(last, dim=-1)
        sess = m.greedy_session(c, first, **kw)
        sess.run(6)
        ref_t, ref_lp = sess.collect()
        assert _hip.decode_engine_status(m._backend._workspace)["engine_launches"] >= 6
        c2, last2 = _prefill(m, prompts, 60)
        sess2 = m.greedy_session(c2, torch.argmax(last2, dim=-1), **kw)
        sess2.logits.fill_(float("nan"))               # what a draw from the never-written buffer would see
        _hip.debug_engine_sabotage(m._backend._workspace, 1)
        sess2.run(6)                                   # step 0 fails its gate, the rest find the workspace poisoned
        torch.cuda.synchronize()
        assert _hip.decode_engine_status(m._backend._workspace)["status"] == 0x700
        assert torch.equal(sess2.buf.tok, first), (sess2.buf.tok.tolist(), first.tolist())   # the input id survived
        got_t, got_lp = sess2.collect()                # re-runs the six steps on the launch path
        assert torch.equal(got_t, ref_t), (got_t[:, 0].tolist(), ref_t[:, 0].tolist())
        assert float((got_lp - ref_lp).abs().max()) < 2e-5
    finally:
        if m._backend._workspace is not None:
            _hip.debug_engine_sabotage(m._backend._workspace, 0)
        _hip.set_decode_engine(True)


def test_residency_gate_failure_writes_nothing_and_steps_are_rerun():
    """An engine launch whose residency census fails (here: sabotaged to wait for one workgroup too many) must leave
    position, rings and samples untouched, poison the workspace (later launches leave at once) and report 0x700;
    GreedySession.collect() then re-runs the missing steps on the launch path: same tokens as an undisturbed run."""
    from mistral_inference import _hip
    m = _model(DENSE, seed=8)
    prompts = _prompts(1, DENSE["vocab_size"], 4)
    ref_t, ref_lp, _ = _session(m, prompts, 12, graph=True)
    try:
        c, last = _prefill(m, prompts, 60)
        first = torch.argmax(last, dim=-1)
        sess = m.greedy_session(c, first, graph=True)
        sess.run(4)
        a_t, _ = sess.collect()
        kv_before = int(c.kv_seqlens[0])
        _hip.debug_engine_sabotage(m._backend._workspace, 1)
        sess.run(7)                       # the first of these fails its gate; the other six find the workspace poisoned
        torch.cuda.synchronize()
        st = _hip.decode_engine_status(m._backend._workspace)
        assert st["status"] == 0x700, st
        assert int(c.kv_seqlens[0]) == kv_before          # nothing advanced
        b_t, _ = sess.collect()           # notices, resets, re-runs the 7 steps on the launch path
        st = _hip.decode_engine_status(m._backend._workspace)
        assert st["status"] == 0 and int(c.kv_seqlens[0]) == kv_before + 7
        got = torch.cat([first[None], a_t, b_t])
        assert torch.equal(got, ref_t), (got[:, 0].tolist(), ref_t[:, 0].tolist())
        assert _hip.set_decode_engine(False) is False     # the session switched the process to the launch path ...
        # ... for the rest of THAT generation: the next session forgets the verdict and probes the device again
        launches = _hip.decode_engine_status(m._backend._workspace)["engine_launches"]
        c2, last2 = _prefill(m, prompts, 60)
        sess2 = m.greedy_session(c2, torch.argmax(last2, dim=-1), graph=True)
        sess2.run(5)
        t2, _ = sess2.collect()
        st = _hip.decode_engine_status(m._backend._workspace)
        assert st["status"] == 0 and st["engine_launches"] >= launches + 5   # back on the persistent engine
        assert torch.equal(t2[:, 0], ref_t[1:6, 0])
        # forward() callers: the flag is raised (and cleared) by raise_if_flagged
        _hip.debug_engine_sabotage(m._backend._workspace, 1)
        m.forward(first, [1], c)
        with pytest.raises(RuntimeError, match="0x700"):
            m._backend.raise_if_flagged()
        assert _hip.decode_engine_status(m._backend._workspace)["status"] == 0
    finally:
        if m._backend._workspace is not None:
            _hip.debug_engine_sabotage(m._backend._workspace, 0)
        _hip.set_decode_engine(True)
