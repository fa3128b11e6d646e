"""The reference's two self-consistency tests (tests/test_generate.py:36-69 and :199-230), restated on the
oracle: greedy-decode then re-score prompt+generation in one prefill (and in chunks of 5) -- the decode
branch, the first-prefill branch and the chunked branch must agree to 5e-4 in fp32."""
import torch

import mistral_oracle as mo

ARGS = mo.OracleArgs(dim=512, n_layers=1, head_dim=128, hidden_dim=2048, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
                     vocab_size=32000)


def _model(**over):
    a = mo.OracleArgs(**{**ARGS.__dict__, **over})
    return mo.OracleModel(a, mo.synth_weights(a, seed=42, dtype=torch.float32))


def test_generation_decode_equals_prefill():
    m = _model()
    enc = [[0, 1, 2, 3, 4, 5, 6, 7], [0, 0, 1, 2], [0, 12, 13, 14], [0, 2, 4, 34]]
    toks, lp_old = mo.generate(enc, m, max_tokens=7)
    enc2 = [e + t for e, t in zip(enc, toks)]
    gen, lp_new = mo.generate(enc2, m, max_tokens=0)
    assert gen == [] and len(toks[0]) == 7
    for a, b in zip(lp_old, lp_new):
        assert all(abs(x - y) < 5e-4 for x, y in zip(a, b))


def test_chunks_equal_one_shot():
    m = _model()
    enc = [[0] + list(range(7)), [0] + list(range(9, 0, -1))]
    toks, lp_old = mo.generate(enc, m, max_tokens=8, max_batch_size=3)
    enc2 = [e + t for e, t in zip(enc, toks)]
    gen, lp_new = mo.generate(enc2, m, max_tokens=0, chunk_size=5, max_batch_size=3)
    assert gen == []
    for a, b in zip(lp_old, lp_new):
        assert all(abs(x - y) < 5e-4 for x, y in zip(a, b))


def test_sliding_window_and_moe_selfconsistency():
    """Cases the reference's tests never reach (coverage gaps)."""
    for over in (dict(sliding_window=4), dict(sliding_window=[4, None], n_layers=2),
                 dict(num_experts=8, num_experts_per_tok=2, hidden_dim=512)):
        m = _model(**over)
        enc = [[0, 1, 2, 3, 4, 5, 6, 7, 8], [0, 5, 1]]
        toks, lp_old = mo.generate(enc, m, max_tokens=6)
        enc2 = [e + t for e, t in zip(enc, toks)]
        for chunk in (None, 8):  # every prompt needs a token in every chunk (generate.py:94)
            gen, lp_new = mo.generate(enc2, m, max_tokens=0, chunk_size=chunk)
            assert gen == []
            for a, b in zip(lp_old, lp_new):
                assert all(abs(x - y) < 5e-4 for x, y in zip(a, b)), (over, chunk)


def test_layer_major_pipeline_of_one_layer_oracles_equals_whole_model():
    """The harness of tests/test_gpu_depth.py: an L-stage pipeline of one-layer OracleModels run LAYER-MAJOR (all
    forwards of stage l before stage l + 1, each stage with its own one-layer cache) gives exactly the whole model's
    logits - teacher-forced schedules make the stages independent of each other's timing."""
    import torch.nn.functional as F
    args = mo.OracleArgs(dim=256, n_layers=3, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
                         vocab_size=300, sliding_window=6)
    w = mo.synth_weights(args, seed=3, dtype=torch.float32)
    ids = torch.randint(0, 300, (14,), generator=torch.Generator().manual_seed(1))
    P, S = 10, 4
    whole = mo.OracleModel(args, w)
    oc = mo.OracleCache(3, 1, 20, 2, 128, 6)
    ref = [whole.forward(ids[:P], [P], oc)] + [whole.forward(ids[P + s:P + s + 1], [1], oc) for s in range(S)]
    h_pre, h_dec = None, [None] * S
    for l in range(3):
        om = mo.OracleModel(args, w, pipeline_rank=l, num_pipeline_ranks=3)
        c = mo.OracleCache(1, 1, 20, 2, 128, 6)
        h_pre = om.forward_partial(ids[:P], [P], c, h_in=h_pre)
        h_dec = [om.forward_partial(ids[P + s:P + s + 1], [1], c, h_in=h_dec[s]) for s in range(S)]
    got = [F.linear(h_pre, w["output.weight"]).float()] + [F.linear(h, w["output.weight"]).float() for h in h_dec]
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
