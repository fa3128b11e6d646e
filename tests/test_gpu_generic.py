"""Storage dtypes other than bf16 and shapes the tuned kernels decline, through `mi_forward_generic` (csrc/generic.hip).

The reference keeps whatever dtype `from_folder(dtype=...)` / `.to(dtype=...)` asks for (transformer.py:303,338) and its own
tests run fp32 models (tests/test_generate.py:51,100).  Parity here is against
  (a) the outputs of the UNMODIFIED reference stored in tests/golden - every fp32 schedule (ragged prefill, sliding window,
      chunks, per-layer windows, MoE, decode) and three fp16 schedules - replayed in the case's OWN dtype: fp32 within 2e-5
      max-abs (no rounding anywhere: what is left is fp32 summation order), fp16 within a few fp16 spacings;
  (b) the CPU oracle in the same dtype;
  (c) the reference's own self-consistency tests (prefill log-probabilities == decode log-probabilities within 5e-4 in fp32,
      tests/test_generate.py:36-69,199-230) through this package's `generate()`;
and for bf16 models whose shape `mi_forward` declines (head_dim 64, top_k = 3, 32 experts) against the bf16 oracle."""
import pytest
import torch

import mistral_oracle as mo
from golden_util import GENERIC_CASES, Case
from hip_util import write_checkpoint

pytestmark = pytest.mark.gpu
F32, F16, BF = torch.float32, torch.float16, torch.bfloat16


def _load(tmp_path, args, weights, dtype, max_batch_size=4):
    from mistral_inference.transformer import Transformer
    folder = write_checkpoint(tmp_path / "ckpt", args, weights)
    model = Transformer.from_folder(folder, max_batch_size=max_batch_size, device="cuda", dtype=dtype)
    assert model.dtype == dtype
    return model


def _replay(model, prompts, tokens, chunk, max_tokens, dtype):
    from mistral_inference.cache import BufferCache
    lens = [len(p) for p in prompts]
    a = model.args
    cache = BufferCache(model.n_local_layers, a.max_batch_size, max(lens) + max_tokens, a.n_kv_heads, a.head_dim, a.sliding_window,
                        device="cuda", dtype=dtype)
    cache.reset()
    chunk = chunk or max(lens)
    pre, dec = [], []
    for s in range(0, max(lens), chunk):
        parts = [p[s:s + chunk] for p in prompts]
        pre.append(model.forward(torch.tensor(sum(parts, []), device="cuda"), [len(p) for p in parts], cache).cpu())
    for step in range(len(tokens[0]) if tokens else 0):
        nxt = torch.tensor([t[step] for t in tokens], device="cuda")
        dec.append(model.forward(nxt, [1] * len(tokens), cache).cpu())
    return pre, dec


def _replay_oracle(args, weights, prompts, tokens, chunk, max_tokens, max_batch_size, dtype):
    model = mo.OracleModel(args, weights)
    lens = [len(p) for p in prompts]
    cache = mo.OracleCache(args.n_layers, max_batch_size, max(lens) + max_tokens, args.n_kv_heads, args.head_dim,
                           args.sliding_window, dtype=dtype)
    chunk = chunk or max(lens)
    pre, dec = [], []
    for s in range(0, max(lens), chunk):
        parts = [p[s:s + chunk] for p in prompts]
        pre.append(model.forward(torch.tensor(sum(parts, []), dtype=torch.long), [len(p) for p in parts], cache))
    for step in range(len(tokens[0]) if tokens else 0):
        dec.append(model.forward(torch.tensor([t[step] for t in tokens], dtype=torch.long), [1] * len(tokens), cache))
    return pre, dec


def _unambiguous_rows(case, trace, n_fwd_rows, schedule, gap_floor):
    """MoE: per forward, the rows whose sequence has not yet hit a router near-tie (k-th vs (k+1)-th logit closer than
    `gap_floor`): from the tie on, which expert runs is a coin flip between two correct implementations."""
    if trace is None:
        return [list(range(n)) for n in n_fwd_rows]
    L, k, B = case.args.n_layers, case.args.num_experts_per_tok, len(schedule[0])
    first = [None] * B
    for f, seqlens in enumerate(schedule):
        for l in range(L):
            srt = torch.sort(trace[f * L + l], dim=1, descending=True).values
            bad = (srt[:, k - 1] - srt[:, k]) <= gap_floor * srt[:, k - 1].abs().clamp(min=1e-3)
            o = 0
            for b, s in enumerate(seqlens):
                if bad[o:o + s].any() and first[b] is None:
                    first[b] = f
                o += s
    keep = []
    for f, seqlens in enumerate(schedule):
        rows, o = [], 0
        for b, s in enumerate(seqlens):
            if first[b] is None or f < first[b]:
                rows += list(range(o, o + s))
            o += s
        keep.append(rows)
    return keep


@pytest.mark.parametrize("name", GENERIC_CASES)
def test_golden_schedules_in_their_own_dtype(name, tmp_path):
    case = Case(name)
    dtype = case.dtype
    w = case.weights()
    model = _load(tmp_path, case.args, w, dtype)
    assert model._backend.plan(model) is not None and model._backend.generic
    toks = case.tokens()
    pre, dec = _replay(model, case.prompts, toks, case.chunk_size, case.max_tokens, dtype)
    mo.ROUTER_TRACE = [] if case.args.num_experts else None
    o_pre, o_dec = _replay_oracle(case.args, w, case.prompts, toks, case.chunk_size, case.max_tokens, case.max_batch_size, dtype)
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    lens = [len(p) for p in case.prompts]
    chunk = case.chunk_size or max(lens)
    schedule = [[len(p[s:s + chunk]) for p in case.prompts] for s in range(0, max(lens), chunk)] + [[1] * len(lens)] * len(dec)
    refs = [case.t[f"prefill_logits.{c}"] for c in range(len(pre))] + [case.t[f"decode_logits.{s}"] for s in range(len(dec))]
    gap = 2.0 ** -18 if dtype == F32 else 2.0 ** -8
    keep = _unambiguous_rows(case, trace, [g.shape[0] for g in pre + dec], schedule, gap)
    # fp32: summation order only.  fp16: logits are fp16 VALUES (|x| < 4 here: spacing 2^-9 .. 2^-8); two correct fp16
    # implementations differ by a spacing or two wherever an fp32 sum lands near a rounding boundary, layer after layer.
    atol = 2e-5 if dtype == F32 else 5e-3   # measured: 1.8e-6 / 2.0e-3 (two fp16 spacings below 4)
    worst_ref, worst_orc, n_rows, tot_rows, exact, elems = 0.0, 0.0, 0, 0, 0, 0
    for f, (got, ref, orc) in enumerate(zip(pre + dec, refs, o_pre + o_dec)):
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert torch.isfinite(got).all()
        r = keep[f]
        tot_rows += got.shape[0]
        n_rows += len(r)
        if r:
            worst_ref = max(worst_ref, (got[r] - ref[r]).abs().max().item())
            worst_orc = max(worst_orc, (got[r] - orc[r]).abs().max().item())
            exact += int((got[r] == ref[r]).sum())
            elems += got[r].numel()
    print(f"\n{name}: max |HIP - reference| {worst_ref:.3e}, max |HIP - oracle| {worst_orc:.3e}, bit-equal to the reference "
          f"{exact / max(elems, 1):.3f} of {elems} logits ({n_rows}/{tot_rows} rows free of router ties)")
    assert n_rows >= 0.5 * tot_rows
    assert worst_ref <= atol, (name, "vs the reference's stored logits", worst_ref)
    assert worst_orc <= atol, (name, "vs the oracle", worst_orc)
    if dtype == F16:
        assert exact >= 0.4 * elems, (name, "bit-equal fraction", exact / elems)  # measured 0.52 .. 0.86


@pytest.mark.parametrize("dtype", [F32, F16])
def test_generate_matches_reference_tokens_and_logprobs(dtype, tmp_path):
    """generate() end to end (chunked prompt logprobs through forward() + log_softmax, the fused sampling session on the
    generic kernels) against the reference's stored tokens and log-probabilities."""
    from mistral_inference.generate import generate
    for name in ("dense_fp32", "swa_chunk_fp32") if dtype == F32 else ("dense_fp16", "swa_chunk_fp16"):
        case = Case(name)
        model = _load(tmp_path / name, case.args, case.weights(), dtype)
        toks, lps = generate(case.prompts, model, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
        ref_toks, ref_lps = case.tokens(), case.logprobs()
        tol = 2e-4 if dtype == F32 else 2e-2
        for b, (mine, ref) in enumerate(zip(toks, ref_toks)):
            n = next((i for i, (x, y) in enumerate(zip(mine, ref)) if x != y), len(ref))
            if dtype == F32:
                assert n == len(ref), (name, b, mine, ref)
            assert n >= 1, (name, b, mine, ref)  # (fp16 greedy paths may fork at a near-tie: agreement up to the fork)
            npl = len(case.prompts[b]) - 1 + n
            err = max(abs(x - y) for x, y in zip(lps[b][:npl], ref_lps[b][:npl]))
            assert err <= tol, (name, b, err)


def test_reference_selfconsistency_fp32(tmp_path):
    """The reference's tests/test_generate.py::test_generation_transformer and ::test_chunks, verbatim in shape and bound:
    fp32, dim 512, 1 layer, vocab 32000; log-probabilities of generated tokens == log-probabilities of the same tokens fed
    back as a prompt (whole, and in chunks of 5) within 5e-4."""
    from mistral_inference.generate import generate
    args = mo.OracleArgs(dim=512, n_layers=1, head_dim=128, hidden_dim=2048, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=32000)
    w = {k: v.float() for k, v in mo.synth_weights(args, seed=42, dtype=F32).items()}
    model = _load(tmp_path, args, w, F32, max_batch_size=4)
    encoded = [[0, 1, 2, 3, 4, 5, 6, 7], [0, 0, 1, 2], [0, 12, 13, 14], [0, 2, 4, 34]]
    toks, lp_old = generate(encoded, model, temperature=0.0, max_tokens=7)
    assert len(toks) == 4 and all(len(t) == 7 for t in toks)
    full = [e + t for e, t in zip(encoded, toks)]
    gen, lp_new = generate(full, model, temperature=0.0, max_tokens=0)
    assert gen == []
    for a, b in zip(lp_old, lp_new):
        assert len(a) == len(b) and all(abs(x - y) < 5e-4 for x, y in zip(a, b)), (a, b)
    gen, lp_chunk = generate(full, model, temperature=0.0, max_tokens=0, chunk_size=5)
    assert gen == []
    for a, b in zip(lp_old, lp_chunk):
        assert len(a) == len(b) and all(abs(x - y) < 5e-4 for x, y in zip(a, b)), (a, b)
    # and the generation itself against the oracle (fp32: identical tokens)
    o_toks, o_lps = mo.generate(encoded, mo.OracleModel(args, w), max_tokens=7, max_batch_size=4)
    assert toks == o_toks
    assert max(abs(x - y) for a, b in zip(lp_old, o_lps) for x, y in zip(a, b)) <= 2e-4


@pytest.mark.parametrize("over,dtype", [
    (dict(head_dim=64, n_heads=4, n_kv_heads=2), BF),                             # head_dim the MFMA attention is not built for
    (dict(head_dim=256, n_heads=2, n_kv_heads=1, dim=512), BF),                   # four elements per attention lane
    (dict(head_dim=96, n_heads=4, n_kv_heads=4), F16),                            # not a multiple of 64: element-wise row loads
    (dict(moe=dict(num_experts=8, num_experts_per_tok=3)), BF),                   # top_k = 3
    (dict(moe=dict(num_experts=32, num_experts_per_tok=2), hidden_dim=256), F32),  # more than 16 experts (fp32: in bf16 every
], ids=["head_dim_64", "head_dim_256", "head_dim_96_fp16", "moe_top3", "moe_32_experts"])  # sequence meets a router near-tie at once)
def test_shapes_the_tuned_kernels_decline(over, dtype, tmp_path):
    """Models `mi_forward` answers with MI_ERR_SHAPE run on the generic kernels instead of raising: ragged prefill + decode
    steps against the oracle in the same dtype (bf16: same bound as the tuned path's golden replays, tests/test_gpu_model.py)."""
    from mistral_inference.transformer import tuned_kernels_take
    p = dict(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=512, sliding_window=8)
    p.update(over)
    args = mo.OracleArgs.from_params(p)
    w = {k: v.to(dtype) for k, v in mo.synth_weights(args, seed=3, dtype=BF).items()}
    model = _load(tmp_path, args, w, dtype)
    assert not tuned_kernels_take(model.args)
    assert model._backend.plan(model) is not None and model._backend.generic
    prompts = [[(3 * i + 1) % 512 for i in range(13)], [5, 6, 7], [(7 * i + 2) % 512 for i in range(9)]]
    toks = [[(11 * s + b) % 512 for s in range(4)] for b in range(3)]
    pre, dec = _replay(model, prompts, toks, None, 4, dtype)
    mo.ROUTER_TRACE = [] if args.num_experts else None
    o_pre, o_dec = _replay_oracle(args, w, prompts, toks, None, 4, 4, dtype)
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None

    class _C:  # (the fields _unambiguous_rows reads)
        pass
    c = _C()
    c.args = args
    schedule = [[len(q) for q in prompts]] + [[1] * 3] * len(dec)
    keep = _unambiguous_rows(c, trace, [g.shape[0] for g in pre + dec], schedule, 2.0 ** -6 if dtype == BF else 2.0 ** -18)
    worst, n = 0.0, 0
    for f, (got, orc) in enumerate(zip(pre + dec, o_pre + o_dec)):
        assert torch.isfinite(got).all()
        if keep[f]:
            worst = max(worst, (got[keep[f]] - orc[keep[f]]).abs().max().item())
            n += len(keep[f])
    assert n >= 0.5 * sum(sum(s) for s in schedule)
    assert worst <= {BF: 4e-2, F16: 5e-3, F32: 2e-5}[dtype], worst


@pytest.mark.parametrize("dtype", [F32, F16])
def test_long_ragged_prompt_in_chunks(dtype, tmp_path):
    """Enough (token, head) blocks for the 4-wave attention launch (the short schedules above all take the 16-wave one), a
    64-slot window that the 300-token prompt wraps several times, a second chunk that reads the ring, contractions of
    400 and 150 rows (m-tiles of the MFMA kernel, ragged), then decode steps - against the oracle in the same dtype."""
    args = mo.OracleArgs(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=8, n_kv_heads=2, norm_eps=1e-5, vocab_size=512,
                         sliding_window=64)
    w = {k: v.to(dtype) for k, v in mo.synth_weights(args, seed=5, dtype=BF).items()}
    model = _load(tmp_path, args, w, dtype)
    prompts = [[(3 * i + 1) % 512 for i in range(300)], [(7 * i + 2) % 512 for i in range(250)]]
    toks = [[(11 * s + b) % 512 for s in range(3)] for b in range(2)]
    pre, dec = _replay(model, prompts, toks, 200, 3, dtype)
    o_pre, o_dec = _replay_oracle(args, w, prompts, toks, 200, 3, 4, dtype)
    worst = max((g - o).abs().max().item() for g, o in zip(pre + dec, o_pre + o_dec))
    assert all(torch.isfinite(g).all() for g in pre + dec)
    assert worst <= (2e-5 if dtype == F32 else 5e-3), worst


def test_nocache_call_fp32(tmp_path):
    """cache=None (transformer_layers.py:165: every token sees every token) against the reference's stored hidden state."""
    case = Case("dense_fp32")
    model = _load(tmp_path, case.args, case.weights(), F32)
    flat = torch.tensor(sum(case.prompts, []), device="cuda")
    h = model.forward_partial(flat, [len(p) for p in case.prompts]).cpu()
    ref = case.t["nocache_hidden"]
    assert (h.float() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


def test_sampling_session_fp32_matches_unfused_loop(tmp_path):
    """temperature 0 through the fused session == forward() + argmax + log_softmax on the same model (tokens identical,
    log-probabilities to fp32 rounding), three sequences, a window that wraps."""
    from mistral_inference.generate import generate
    args = mo.OracleArgs(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=512,
                         sliding_window=8)
    w = {k: v.float() for k, v in mo.synth_weights(args, seed=9, dtype=BF).items()}
    model = _load(tmp_path, args, w, F32)
    prompts = [[1, 5, 9, 200, 17, 3, 44, 8, 90, 11, 12], [7, 300, 2], [11, 12, 13, 14, 15]]
    toks, lps = generate(prompts, model, max_tokens=12, temperature=0.0)
    model.fused_greedy = False
    toks2, lps2 = generate(prompts, model, max_tokens=12, temperature=0.0)
    assert toks == toks2
    assert max(abs(x - y) for a, b in zip(lps, lps2) for x, y in zip(a, b)) <= 1e-5
    o_toks, _ = mo.generate(prompts, mo.OracleModel(args, w), max_tokens=12, max_batch_size=4)
    assert toks == o_toks


def test_fma_tile_kernel_on_the_same_schedules():
    """MI_GENERIC_MFMA=0 sends every multi-row contraction to the 64 x 64 fp32-FMA tile kernel (what rows that are not 16-byte
    aligned take): one dense and one MoE golden schedule per dtype in a process of their own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MI_GENERIC_MFMA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_generic.py"), "-q", "-x", "-k",
                        "own_dtype and (dense_fp32 or dense_fp16 or moe_fp32 or moe_fp16)"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(here))
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", range(14))
def test_fuzz_whole_models(seed, tmp_path):
    """Seeded random models nobody picked by hand - dims, head layouts (MHA / GQA / MQA), head dims 32 .. 128 including ones
    that are not a multiple of 64, dense and MoE (top-1 / 2 / 3 of 4 .. 12), global / per-layer / no sliding window, vocabularies
    that are not a tile multiple, ragged batches, chunked prompts, decode steps - in a random storage dtype against the oracle in
    that dtype (MoE in fp32 only: no router near-ties to argue about)."""
    import random
    rng = random.Random(1000 + seed)
    Dh = rng.choice([32, 64, 80, 96, 128])
    Hkv = rng.choice([1, 2, 4])
    H = Hkv * rng.choice([1, 2, 3, 4])
    moe = rng.random() < 0.4
    dtype = F32 if moe else rng.choice([F32, F16, BF])
    L = rng.randint(1, 3)
    p = dict(dim=8 * rng.randint(8, 48), n_layers=L, head_dim=Dh, hidden_dim=8 * rng.randint(8, 80), n_heads=H, n_kv_heads=Hkv,
             norm_eps=1e-5, vocab_size=rng.randint(100, 900), rope_theta=rng.choice([1e4, 1e6]))
    if moe:
        E = rng.choice([4, 6, 8, 12])
        p["moe"] = dict(num_experts=E, num_experts_per_tok=rng.randint(1, 3))
    win = rng.choice(["none", "one", "list"])
    if win == "one":
        p["sliding_window"] = rng.randint(3, 24)
    elif win == "list":
        p["sliding_window"] = [rng.choice([None, rng.randint(3, 24)]) for _ in range(L)]
    args = mo.OracleArgs.from_params(p)
    w = {k: v.to(dtype) for k, v in mo.synth_weights(args, seed=seed, dtype=BF).items()}
    model = _load(tmp_path, args, w, dtype)
    assert model._backend.plan(model) is not None and (model._backend.generic or (dtype == BF and Dh == 128))
    B = rng.randint(1, 4)
    prompts = [[rng.randrange(p["vocab_size"]) for _ in range(rng.randint(1, 40))] for _ in range(B)]
    chunk = rng.choice([None, None, 7, 16])
    if chunk is not None:  # (every prompt needs a token in every chunk, reference generate.py:94)
        n_chunks = -(-max(len(q) for q in prompts) // chunk)
        prompts = [q + [rng.randrange(p["vocab_size"]) for _ in range(max(0, (n_chunks - 1) * chunk + 1 - len(q)))] for q in prompts]
    n_dec = 3
    toks = [[rng.randrange(p["vocab_size"]) for _ in range(n_dec)] for _ in range(B)]
    pre, dec = _replay(model, prompts, toks, chunk, n_dec, dtype)
    o_pre, o_dec = _replay_oracle(args, w, prompts, toks, chunk, n_dec, 4, dtype)
    tol = {F32: 2e-5, F16: 6e-3, BF: 5e-2}[dtype]
    for f, (g, o) in enumerate(zip(pre + dec, o_pre + o_dec)):
        assert g.shape == o.shape and torch.isfinite(g).all(), (seed, p, f)
        err = (g - o).abs().max().item()
        assert err <= tol * max(1.0, o.abs().max().item()), (seed, p, str(dtype), chunk, [len(q) for q in prompts], f, err)
