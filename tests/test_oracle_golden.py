"""The oracle restatement vs outputs of the UNMODIFIED reference (tests/golden, oracle/make_golden.py)."""
import pytest
import torch

import mistral_oracle as mo
from golden_util import CASES, Case


def _replay(case: Case):
    """Re-run the golden's generate() schedule with the oracle, teacher-forcing the reference's tokens."""
    model = mo.OracleModel(case.args, case.weights())
    lens = [len(p) for p in case.prompts]
    cache = mo.OracleCache(case.args.n_layers, case.max_batch_size, max(lens) + case.max_tokens,
                           case.args.n_kv_heads, case.args.head_dim, case.args.sliding_window, dtype=case.dtype)
    chunk = case.chunk_size or max(lens)
    outs = []
    hidden = []
    for c, s in enumerate(range(0, max(lens), chunk)):
        parts = [p[s:s + chunk] for p in case.prompts]
        ids = torch.tensor(sum(parts, []), dtype=torch.long)
        if c == 0:
            h = model.forward_partial(ids, [len(p) for p in parts], cache, collect=hidden)
            import torch.nn.functional as F
            outs.append(F.linear(h, model.w["output.weight"]).float())
        else:
            outs.append(model.forward(ids, [len(p) for p in parts], cache))
    toks = case.tokens()
    dec = []
    for step in range(case.n_decode()):
        nxt = torch.tensor([t[step] for t in toks], dtype=torch.long)
        dec.append(model.forward(nxt, [1] * len(toks), cache))
    return outs, hidden, dec


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(name):
    case = Case(name)
    atol, _ = case.tol()
    outs, hidden, dec = _replay(case)
    for c, o in enumerate(outs):
        ref = case.t[f"prefill_logits.{c}"]
        assert o.shape == ref.shape
        assert (o - ref).abs().max().item() <= atol, (name, "prefill", c, (o - ref).abs().max().item())
    for i, h in enumerate(hidden):
        ref = case.t[f"prefill_hidden.{i}"]
        scale = max(1.0, ref.abs().max().item())
        assert (h.float() - ref).abs().max().item() <= atol * scale, (name, "hidden", i)
    for s, o in enumerate(dec):
        ref = case.t[f"decode_logits.{s}"]
        assert (o - ref).abs().max().item() <= atol, (name, "decode", s, (o - ref).abs().max().item())


@pytest.mark.parametrize("name", CASES)
def test_generate_matches_reference(name):
    case = Case(name)
    _, lp_tol = case.tol()
    model = mo.OracleModel(case.args, case.weights())
    toks, lps = mo.generate(case.prompts, model, max_tokens=case.max_tokens, max_batch_size=case.max_batch_size,
                            chunk_size=case.chunk_size)
    ref_lps = case.logprobs()
    if case.dtype == torch.float32:
        assert toks == case.tokens()
        for a, b in zip(lps, ref_lps):
            assert len(a) == len(b)
            assert max(abs(x - y) for x, y in zip(a, b)) <= lp_tol
    else:
        # bf16 greedy paths may legitimately fork at a near-tie; require agreement up to the first fork
        for b, (mine, ref) in enumerate(zip(toks, case.tokens())):
            n = next((i for i, (x, y) in enumerate(zip(mine, ref)) if x != y), len(ref))
            assert n >= 1, (name, b, mine, ref)
            npl = len(case.prompts[b]) - 1 + n
            assert max(abs(x - y) for x, y in zip(lps[b][:npl], ref_lps[b][:npl])) <= lp_tol


@pytest.mark.parametrize("name", ["dense_fp32", "dense_bf16"])
def test_nocache_forward_partial(name):
    """cache=None: attention is unmasked across all concatenated tokens (transformer_layers.py:165)."""
    case = Case(name)
    model = mo.OracleModel(case.args, case.weights())
    flat = torch.tensor(sum(case.prompts, []), dtype=torch.long)
    h = model.forward_partial(flat, [len(p) for p in case.prompts], None)
    ref = case.t["nocache_hidden"]
    atol, _ = case.tol()
    assert (h.float() - ref).abs().max().item() <= atol * max(1.0, ref.abs().max().item())
