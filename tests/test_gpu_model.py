"""End-to-end parity on the GPU: `Transformer.from_folder` + `generate()` through libmistral_hip against
(a) outputs of the unmodified reference (tests/golden, bf16 cases) and (b) the CPU oracle in bf16."""
import pytest
import torch

import mistral_oracle as mo
from golden_util import GPU_CASES as CASES, Case
from hip_util import bf16_ulp_close, write_checkpoint

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
LOGIT_ATOL = 4e-2   # bf16 noise floor, see golden_util.Case.tol and DESIGN.md section 5


def _load(tmp_path, args, weights, max_batch_size=4):
    from mistral_inference.transformer import Transformer
    folder = write_checkpoint(tmp_path / "ckpt", args, weights)
    return Transformer.from_folder(folder, max_batch_size=max_batch_size, device="cuda", dtype=BF)


def _replay_hip(model, case: Case, prompts, tokens, chunk):
    from mistral_inference.cache import BufferCache
    lens = [len(p) for p in prompts]
    a = model.args
    cache = BufferCache(model.n_local_layers, a.max_batch_size, max(lens) + case.max_tokens, a.n_kv_heads, a.head_dim,
                        a.sliding_window, device="cuda", dtype=BF)
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


def _ambiguous_from(trace, schedule, n_layers, top_k, B):
    """First forward index at which each sequence hits a router near-tie (gap between the k-th and (k+1)-th
    bf16 logit <= 2 ulp): from there on its outputs legitimately depend on tie-breaking."""
    first = [None] * B
    for f, seqlens in enumerate(schedule):
        for l in range(n_layers):
            lg = trace[f * n_layers + l]
            srt = torch.sort(lg, dim=1, descending=True).values
            gap = srt[:, top_k - 1] - srt[:, top_k]
            ulp = srt[:, top_k - 1].abs().clamp(min=1e-3) * 2.0 ** -7
            bad = gap <= 2 * ulp
            o = 0
            for b, s in enumerate(seqlens):
                if bad[o:o + s].any() and first[b] is None:
                    first[b] = f
                o += s
    return first


def _replay_oracle(case: Case, weights, tokens):
    model = mo.OracleModel(case.args, weights)
    lens = [len(p) for p in case.prompts]
    cache = mo.OracleCache(case.args.n_layers, case.max_batch_size, max(lens) + case.max_tokens, case.args.n_kv_heads,
                           case.args.head_dim, case.args.sliding_window, dtype=BF)
    chunk = case.chunk_size or max(lens)
    pre, dec = [], []
    for s in range(0, max(lens), chunk):
        parts = [p[s:s + chunk] for p in case.prompts]
        pre.append(model.forward(torch.tensor(sum(parts, []), dtype=torch.long), [len(p) for p in parts], cache))
    for step in range(len(tokens[0])):
        dec.append(model.forward(torch.tensor([t[step] for t in tokens], dtype=torch.long), [1] * len(tokens), cache))
    return pre, dec


@pytest.mark.parametrize("name", CASES)
def test_logits_vs_reference_and_oracle(name, tmp_path):
    """Teacher-forced replay of each golden schedule (ragged prefill, chunks, window wrap, MoE, decode)."""
    case = Case(name)
    w = {k: v.to(BF) for k, v in mo.synth_weights(case.args, seed=case.meta["seed"], dtype=BF).items()}
    model = _load(tmp_path, case.args, w)
    toks = case.tokens()
    pre, dec = _replay_hip(model, case, case.prompts, toks, case.chunk_size)
    mo.ROUTER_TRACE = [] if case.args.num_experts else None
    o_pre, o_dec = _replay_oracle(case, w, toks)
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    lens = [len(p) for p in case.prompts]
    chunk = case.chunk_size or max(lens)
    schedule = [[len(p[s:s + chunk]) for p in case.prompts] for s in range(0, max(lens), chunk)]
    schedule += [[1] * len(lens)] * len(dec)
    amb = [None] * len(lens)
    if trace is not None:
        amb = _ambiguous_from(trace, schedule, case.args.n_layers, case.args.num_experts_per_tok, len(lens))

    def rows(f):  # rows of forward f that are free of tie-breaking ambiguity
        keep, o = [], 0
        for b, s in enumerate(schedule[f]):
            if amb[b] is None or f < amb[b]:
                keep += list(range(o, o + s))
            o += s
        return keep

    worst, compared = 0.0, 0
    within, elems, abs_sum = 0, 0, 0.0  # BASELINE.json's "within 1e-2 max-abs": see the note below
    exact, in1, in2, in3 = 0, 0, 0, 0   # ... and in units of the reference value's own bf16 spacing
    refs = [case.t.get(f"prefill_logits.{c}") for c in range(len(pre))] + \
           [case.t.get(f"decode_logits.{s}") for s in range(len(dec))]
    for f, (got, ref) in enumerate(zip(pre + dec, o_pre + o_dec)):
        assert got.shape == ref.shape and got.dtype == torch.float32
        r = rows(f)
        compared += len(r)
        if r:
            worst = max(worst, (got[r] - ref[r]).abs().max().item())
            if case.dtype == BF:  # stored outputs of the unmodified reference
                d = (got[r] - refs[f][r]).abs()
                assert d.max().item() <= LOGIT_ATOL, (name, "vs reference", f)
                within += int((d <= 1e-2).sum())
                elems += d.numel()
                abs_sum += float(d.sum())
                # bf16 spacing at the reference value; below |x| = 1 the error is inherited from the hidden state, not from
                # the logit's own rounding, so the spacing of [1, 2) is the floor (7.8e-3)
                ulp = refs[f][r].abs().clamp(min=1.0) * 2.0 ** -7
                exact += int((d == 0).sum())
                in1 += int((d <= 1.0 * ulp + 1e-7).sum())
                in2 += int((d <= 2.0 * ulp + 1e-7).sum())
                in3 += int((d <= 3.0 * ulp + 1e-7).sum())
    total = sum(sum(s) for s in schedule)
    assert compared >= 0.5 * total, (name, "too many tie-ambiguous rows", compared, total)
    assert worst <= LOGIT_ATOL, (name, "vs bf16 oracle", worst)
    if elems:
        # The north star asks for logits within 1e-2 max-abs of the reference.  Logits are bf16 VALUES (the LM head
        # rounds to bf16 before .float(), transformer.py:235-242): for |x| in [1, 2) two neighbouring bf16 numbers are
        # 7.8e-3 apart and for |x| in [2, 4) 1.56e-2, so ANY two bf16 implementations whose fp32 sums differ in the last
        # bit differ by more than 1e-2 on some elements (the reference does so against itself when only its CPU thread
        # count changes).  What is asserted: max-abs within the 2-3 ulp bound above, at least 97 %
        # of all logits within 1e-2, and a mean error an order of magnitude below it.
        print(f"\n{name}: vs the reference's stored logits - bit-exact {exact / elems:.3f}, within 1 / 2 / 3 bf16 ulp(ref) "
              f"{in1 / elems:.4f} / {in2 / elems:.4f} / {in3 / elems:.4f}, within 1e-2 {within / elems:.4f}, mean |d| {abs_sum / elems:.5f}")
        assert in2 >= 0.999 * elems, (name, "fraction of logits within 2 bf16 ulp of the reference value", in2 / elems)
        assert in3 == elems, (name, "a logit further than 3 bf16 ulp from the reference value", in3 / elems)
        assert exact >= 0.25 * elems, (name, "bit-exact fraction", exact / elems)
        assert within >= 0.97 * elems, (name, "fraction of logits within 1e-2 of the reference", within / elems)
        assert abs_sum / elems <= 2.5e-3, (name, "mean abs logit error vs the reference", abs_sum / elems)


@pytest.mark.parametrize("name", [c for c in CASES if c.endswith("bf16")])
def test_generate_vs_reference(name, tmp_path):
    from mistral_inference.generate import generate
    case = Case(name)
    w = case.weights()
    model = _load(tmp_path, case.args, w)
    toks, lps = generate(case.prompts, model, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
    ref_toks, ref_lps = case.tokens(), case.logprobs()
    assert len(toks) == len(ref_toks)
    agree = 0
    for b, (mine, ref) in enumerate(zip(toks, ref_toks)):
        n = next((i for i, (x, y) in enumerate(zip(mine, ref)) if x != y), len(ref))
        agree += n
        assert n >= 1, (name, b, mine, ref)
        npl = len(case.prompts[b]) - 1 + n
        assert len(lps[b]) == len(ref_lps[b])
        assert max(abs(x - y) for x, y in zip(lps[b][:npl], ref_lps[b][:npl])) <= 6e-2, (name, b)
    assert agree >= 0.6 * sum(len(t) for t in ref_toks), (name, agree)


def test_reference_selfconsistency_decode_vs_prefill(tmp_path):
    """The reference's own test (tests/test_generate.py:36-69): greedy-decode, then re-score prompt+generation in
    one prefill with max_tokens=0 -- decode path (ring + GEMV kernels) == prefill path (MFMA kernels)."""
    from mistral_inference.generate import generate
    args = mo.OracleArgs(dim=512, n_layers=1, head_dim=128, hidden_dim=2048, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
                         vocab_size=32000)
    model = _load(tmp_path, args, mo.synth_weights(args, seed=42))
    enc = [[0, 1, 2, 3, 4, 5, 6, 7], [0, 0, 1, 2], [0, 12, 13, 14], [0, 2, 4, 34]]
    toks, lp_old = generate(enc, model, temperature=0.0, max_tokens=7)
    enc2 = [e + t for e, t in zip(enc, toks)]
    gen, lp_new = generate(enc2, model, temperature=0.0, max_tokens=0)
    assert gen == []
    worst = max(abs(x - y) for a, b in zip(lp_old, lp_new) for x, y in zip(a, b))
    assert worst < 8e-2, worst  # reference asserts 5e-4 in fp32; this is bf16 storage (ulp 7.8e-3 at 1.0)
    # chunked re-score (tests/test_generate.py:199-230)
    gen, lp_chunk = generate(enc2[:3], model, temperature=0.0, max_tokens=0, chunk_size=5)
    assert gen == []
    worst = max(abs(x - y) for a, b in zip(lp_old[:3], lp_chunk) for x, y in zip(a, b))
    assert worst < 8e-2, worst


def test_7b_dims_two_layers_vs_oracle(tmp_path):
    """BASELINE.json configs[0]: Mistral-7B-v0.3 dimensions, 2 layers, 32-token prefill + 16 greedy tokens."""
    from mistral_inference.generate import generate
    args = mo.OracleArgs(dim=4096, n_layers=2, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                         vocab_size=32768, rope_theta=1e6)
    w = mo.synth_weights(args, seed=42)
    prompt = torch.randint(0, args.vocab_size, (32,), generator=torch.Generator().manual_seed(0)).tolist()
    model = _load(tmp_path, args, w, max_batch_size=1)
    toks, lps = generate([prompt], model, max_tokens=16, temperature=0.0)
    o_toks, o_lps = mo.generate([prompt], mo.OracleModel(args, w), max_tokens=16)
    n = next((i for i, (x, y) in enumerate(zip(toks[0], o_toks[0])) if x != y), 16)
    assert n >= 8, (toks, o_toks)
    m = 31 + n
    assert max(abs(x - y) for x, y in zip(lps[0][:m], o_lps[0][:m])) < 6e-2
    # logits of the prefill, all 32 rows
    from mistral_inference.cache import BufferCache
    cache = BufferCache(2, 1, 64, 8, 128, None, device="cuda", dtype=BF)
    got = model.forward(torch.tensor(prompt, device="cuda"), [32], cache).cpu()
    ocache = mo.OracleCache(2, 1, 64, 8, 128, None, dtype=BF)
    ref = mo.OracleModel(args, w).forward(torch.tensor(prompt), [32], ocache)
    err = (got - ref).abs().max().item()
    assert err <= 3e-2, err


def test_forward_partial_nocache(tmp_path):
    case = Case("dense_bf16")
    model = _load(tmp_path, case.args, case.weights())
    flat = torch.tensor(sum(case.prompts, []), device="cuda")
    h = model.forward_partial(flat, [len(p) for p in case.prompts]).float().cpu()
    ref = case.t["nocache_hidden"]
    assert (h - ref).abs().max().item() <= 4e-2 * max(1.0, ref.abs().max().item())


def test_fails_loudly_off_device(tmp_path):
    from mistral_inference.transformer import Transformer
    case = Case("dense_bf16")
    folder = write_checkpoint(tmp_path / "c", case.args, case.weights())
    m = Transformer.from_folder(folder, max_batch_size=2, device="cpu", dtype=BF)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward(torch.tensor([1, 2, 3]), [3])
    # storage dtypes: bf16 (tuned kernels), fp16 and fp32 (generic kernels, tests/test_gpu_generic.py) run; anything else is
    # refused at the first forward instead of being converted behind the caller's back
    m64 = Transformer.from_folder(folder, max_batch_size=2, device="cuda", dtype=torch.float64)
    with pytest.raises(RuntimeError, match="storage dtype"):
        m64.forward(torch.tensor([1, 2, 3], device="cuda"), [3])


def test_decode_batch_larger_than_gemv_limit(tmp_path):
    """12 sequences decoding together: T = 12 > 8 takes the MFMA GEMM kernels with the decode attention branch."""
    from mistral_inference.cache import BufferCache
    args = mo.OracleArgs(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
                         vocab_size=512, sliding_window=8)
    w = mo.synth_weights(args, seed=11)
    model = _load(tmp_path, args, w, max_batch_size=12)
    prompts = [[(3 * i + 7 * b + 1) % 512 for i in range(3 + (b % 5))] for b in range(12)]
    lens = [len(p) for p in prompts]
    cache = BufferCache(2, 12, 32, 2, 128, 8, device="cuda", dtype=BF)
    ocache = mo.OracleCache(2, 12, 32, 2, 128, 8, dtype=BF)
    om = mo.OracleModel(args, w)
    flat = sum(prompts, [])
    got = model.forward(torch.tensor(flat, device="cuda"), lens, cache).cpu()
    ref = om.forward(torch.tensor(flat), lens, ocache)
    assert (got - ref).abs().max().item() <= LOGIT_ATOL
    for step in range(10):  # crosses the W=8 ring wrap for every sequence
        nxt = torch.tensor([(5 * b + step) % 512 for b in range(12)])
        got = model.forward(nxt.cuda(), [1] * 12, cache).cpu()
        ref = om.forward(nxt, [1] * 12, ocache)
        assert (got - ref).abs().max().item() <= LOGIT_ATOL, step
    assert cache.kv_seqlens.tolist() == [n + 10 for n in lens]


def test_softmax_fp32_false_and_module_level_block(tmp_path):
    from mistral_inference.transformer import Transformer
    case = Case("dense_bf16")
    folder = write_checkpoint(tmp_path / "c", case.args, case.weights())
    m = Transformer.from_folder(folder, max_batch_size=4, device="cuda", dtype=BF, softmax_fp32=False)
    m32 = Transformer.from_folder(folder, max_batch_size=4, device="cuda", dtype=BF)
    ids = torch.tensor(sum(case.prompts, []), device="cuda")
    lens = [len(p) for p in case.prompts]
    a, b = m.forward(ids, lens), m32.forward(ids, lens)
    assert a.dtype == BF and b.dtype == torch.float32 and torch.equal(a.float(), b)  # widening is exact
    # module-level TransformerBlock.forward (cache=None) == the runner's first layer
    blk = m32.layers["0"]
    h0 = m32.tok_embeddings.weight[ids]
    pos = torch.cat([torch.arange(n) for n in lens]).cuda()
    out = blk(h0, m32.freqs_cis[pos]).float().cpu()
    om = mo.OracleModel(case.args, case.weights())
    col = []
    om.forward_partial(ids.cpu(), lens, None, collect=col)
    assert (out - col[0].float()).abs().max().item() <= 4e-2 * max(1.0, col[0].float().abs().max().item())


def test_load_lora_merges_like_the_reference(tmp_path):
    """Transformer.load_lora (reference lora.py:92-139, merged form): W <- W + (B @ A) * scaling in the model dtype for
    every linear that has adapter keys; ranks that are not a multiple of 8 exercise the zero padding of the GEMM's K.
    Checked on the merged weights (1 bf16 ulp: fp32 summation order of B @ A) and on logits against the oracle run on
    weights merged the reference's way on the CPU."""
    from safetensors.torch import save_file
    args = mo.OracleArgs(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, vocab_size=320,
                         norm_eps=1e-5, rope_theta=1e6, sliding_window=None)
    weights = mo.synth_weights(args, seed=7)
    model = _load(tmp_path, args, weights)
    g = torch.Generator().manual_seed(11)
    lora, merged = {}, dict(weights)
    for name, r in (("layers.0.attention.wq", 8), ("layers.0.feed_forward.w2", 5), ("layers.1.attention.wo", 16),
                    ("layers.1.feed_forward.w1", 3)):
        w = weights[name + ".weight"]
        a = (torch.randn(r, w.shape[1], generator=g) * 0.05).to(BF)
        b = (torch.randn(w.shape[0], r, generator=g) * 0.05).to(BF)
        lora[name + ".lora_A.weight"], lora[name + ".lora_B.weight"] = a, b
        merged[name + ".weight"] = w + (b @ a) * 2.0  # lora.py:131-135, bf16 tensors
    path = tmp_path / "lora.safetensors"
    save_file(lora, str(path))
    model.load_lora(path)
    sd = model.state_dict()
    for name in merged:
        ok, err = bf16_ulp_close(sd[name].cpu(), merged[name], ulps=1.0)
        assert ok, (name, err)
    prompt = torch.randint(0, args.vocab_size, (17,), generator=g)
    got = model.forward(prompt.cuda(), [17]).cpu()
    ref = mo.OracleModel(args, merged).forward(prompt, [17], None)
    assert float((got - ref).abs().max()) < LOGIT_ATOL
    # contract errors of the reference loader
    with pytest.raises(AssertionError):
        model._load_lora_state_dict({"layers.0.attention.wq.lora_A.weight": torch.zeros(8, 256)})  # fp32 != bf16
    with pytest.raises(AssertionError):
        model._load_lora_state_dict({"layers.0.attention.wq.weight": torch.zeros(8, 256, dtype=BF)})  # not a lora key


def test_moe_long_prefill_takes_256_row_tiles(tmp_path):
    """T * top_k >= 512 * E switches the token-grouped expert GEMMs to 256-row m-tiles (gemm256.hip): one MoE layer,
    1200-token prompt, every token's logits against the oracle.  With ONE layer a router near-tie only affects its own
    token (routing happens after the attention), so exactly those tokens are excluded."""
    args = mo.OracleArgs(dim=256, n_layers=1, head_dim=128, hidden_dim=512, n_heads=2, n_kv_heads=1, vocab_size=320,
                         norm_eps=1e-5, rope_theta=1e6, num_experts=4, num_experts_per_tok=2)
    w = mo.synth_weights(args, seed=21)
    model = _load(tmp_path, args, w, max_batch_size=1)
    T = 1200
    ids = torch.randint(0, args.vocab_size, (T,), generator=torch.Generator().manual_seed(22))
    got = model.forward(ids.cuda(), [T]).cpu()
    mo.ROUTER_TRACE = []
    ref = mo.OracleModel(args, w).forward(ids, [T], None)
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    srt = torch.sort(trace[0], dim=1, descending=True).values
    gap = srt[:, 1] - srt[:, 2]
    clear = gap > 2 * srt[:, 1].abs().clamp(min=1e-3) * 2.0 ** -7
    assert int(clear.sum()) >= 0.8 * T
    assert float((got[clear] - ref[clear]).abs().max()) <= LOGIT_ATOL


def test_mixtral_8x22b_dims_one_layer_vs_oracle():
    """BASELINE.json configs[4] shapes (dim 6144, 48 q heads over 8 kv heads = GQA ratio 6, hidden 16384, 8 experts top-2),
    ONE layer (4.9 GB), small vocabulary: prefill logits of a 40-token prompt and 3 teacher-forced decode steps against
    the bf16 oracle.  Weights are generated on the device and copied to the host for the oracle (no 5 GB checkpoint on
    disk).  One MoE layer: a router near-tie only touches its own token, those tokens are excluded."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    p = dict(dim=6144, n_layers=1, head_dim=128, hidden_dim=16384, n_heads=48, n_kv_heads=8, norm_eps=1e-5,
             vocab_size=2048, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2))
    ta = TransformerArgs.from_dict(p)
    ta.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(ta)
    model = model.to(BF).to_empty(device="cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("norm.weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g, device="cuda"))
            elif name.startswith("tok_embeddings"):
                prm.copy_(torch.randn(prm.shape, generator=g, device="cuda"))
            else:
                prm.copy_((torch.rand(prm.shape, generator=g, device="cuda") * 2 - 1) / prm.shape[1] ** 0.5)
    model._weights_changed()
    model.eval()
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    oargs = mo.OracleArgs.from_params(p)
    T, steps = 40, 3
    ids = torch.randint(0, 2048, (T + steps,), generator=torch.Generator().manual_seed(8))
    cache = BufferCache(1, 1, T + steps + 2, 8, 128, None, device="cuda", dtype=BF)
    cache.reset()
    got = [model.forward(ids[:T].cuda(), [T], cache).cpu()]
    got += [model.forward(ids[T + i:T + i + 1].cuda(), [1], cache).cpu() for i in range(steps)]
    om = mo.OracleModel(oargs, w)
    oc = mo.OracleCache(1, 1, T + steps + 2, 8, 128, None, dtype=BF)
    mo.ROUTER_TRACE = []
    ref = [om.forward(ids[:T], [T], oc)] + [om.forward(ids[T + i:T + i + 1], [1], oc) for i in range(steps)]
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    kept = 0
    for gl, rl, lg in zip(got, ref, trace):
        srt = torch.sort(lg, dim=1, descending=True).values
        clear = (srt[:, 1] - srt[:, 2]) > 2 * srt[:, 1].abs().clamp(min=1e-3) * 2.0 ** -7
        kept += int(clear.sum())
        if clear.any():
            assert float((gl[clear] - rl[clear]).abs().max()) <= LOGIT_ATOL
    assert kept >= 0.8 * (T + steps)


def test_nemo_12b_dims_one_layer_vs_oracle(tmp_path):
    """BASELINE.json configs[2] shapes: dim 5120 with 32 heads of 128 (n_heads * head_dim = 4096 != dim), hidden 14336;
    one layer, small vocabulary: 300-token prefill (the 256-tile GEMMs, whose 5120 output columns leave a partly filled
    last round) and 3 decode steps against the bf16 oracle."""
    args = mo.OracleArgs(dim=5120, n_layers=1, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                         vocab_size=1024, rope_theta=1e6)
    w = mo.synth_weights(args, seed=31)
    model = _load(tmp_path, args, w, max_batch_size=1)
    from mistral_inference.cache import BufferCache
    T, steps = 300, 3
    ids = torch.randint(0, args.vocab_size, (T + steps,), generator=torch.Generator().manual_seed(32))
    cache = BufferCache(1, 1, T + steps + 2, 8, 128, None, device="cuda", dtype=BF)
    cache.reset()
    got = [model.forward(ids[:T].cuda(), [T], cache).cpu()]
    got += [model.forward(ids[T + i:T + i + 1].cuda(), [1], cache).cpu() for i in range(steps)]
    om = mo.OracleModel(args, w)
    oc = mo.OracleCache(1, 1, T + steps + 2, 8, 128, None, dtype=BF)
    ref = [om.forward(ids[:T], [T], oc)] + [om.forward(ids[T + i:T + i + 1], [1], oc) for i in range(steps)]
    for gl, rl in zip(got, ref):
        assert float((gl - rl).abs().max()) <= LOGIT_ATOL


def test_module_level_moe_block_vs_oracle(tmp_path):
    """TransformerBlock.forward with a MoeLayer driven module by module (router kernel + per-expert fused FFN + ordered
    bf16 accumulation) against the oracle's first layer; tokens whose top-2 pick is a near-tie are excluded."""
    from mistral_inference.transformer import Transformer
    case = Case("moe_bf16")
    folder = write_checkpoint(tmp_path / "c", case.args, case.weights())
    m = Transformer.from_folder(folder, max_batch_size=4, device="cuda", dtype=BF)
    ids = torch.tensor(sum(case.prompts, []), device="cuda")
    lens = [len(p) for p in case.prompts]
    pos = torch.cat([torch.arange(n) for n in lens]).cuda()
    with torch.no_grad():
        out = m.layers["0"](m.tok_embeddings.weight[ids], m.freqs_cis[pos]).float().cpu()
    om = mo.OracleModel(case.args, case.weights())
    col = []
    mo.ROUTER_TRACE = []
    om.forward_partial(ids.cpu(), lens, None, collect=col)
    trace, mo.ROUTER_TRACE = mo.ROUTER_TRACE, None
    srt = torch.sort(trace[0], dim=1, descending=True).values
    k = case.args.num_experts_per_tok
    clear = (srt[:, k - 1] - srt[:, k]) > 2 * srt[:, k - 1].abs().clamp(min=1e-3) * 2.0 ** -7
    assert int(clear.sum()) >= 0.7 * len(ids)
    ref = col[0].float()
    assert float((out[clear] - ref[clear]).abs().max()) <= 4e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("name", ["dense_bf16", "swa_bf16"])
def test_module_level_decode_with_cache(name, tmp_path):
    """The PUBLIC layer-by-layer route at decode (reference transformer.py:196-211 written out by a caller):
    cache.get_input_metadata -> cache.get_view -> TransformerBlock.forward(h, freqs_cis[positions], view) -> norm ->
    output, after a prefill through the runner.  Decode metadata comes from the cache's host mirror (the runner's
    decode-prep kernel never ran for these views); rings, positions and logits must match the reference's stored decode
    logits like the runner's do."""
    from mistral_inference.cache import BufferCache
    case = Case(name)
    w = {k: v.to(BF) for k, v in mo.synth_weights(case.args, seed=case.meta["seed"], dtype=BF).items()}
    m = _load(tmp_path, case.args, w)
    a = m.args
    lens = [len(p) for p in case.prompts]
    B = len(lens)
    cache = BufferCache(m.n_local_layers, a.max_batch_size, max(lens) + case.max_tokens, a.n_kv_heads, a.head_dim,
                        a.sliding_window, device="cuda", dtype=BF)
    cache.reset()
    m.forward(torch.tensor(sum(case.prompts, []), device="cuda"), lens, cache)
    toks = case.tokens()
    for step in range(3):
        nxt = torch.tensor([t[step] for t in toks], device="cuda")
        md = cache.get_input_metadata([1] * B)
        assert not md[0].prefill and md[0].positions.tolist() == [n + step for n in lens]
        h = m.tok_embeddings.weight[nxt]
        for li, blk in enumerate(m.layers.values()):
            h = blk(h, m.freqs_cis[md[li].positions], cache.get_view(li, md[li]))
        cache.update_seqlens([1] * B)
        logits = torch.nn.functional.linear(m.norm(h), m.output.weight).float().cpu()
        ref = case.t[f"decode_logits.{step}"]
        assert float((logits - ref).abs().max()) <= LOGIT_ATOL, (step, float((logits - ref).abs().max()))
    assert cache.kv_seqlens.tolist() == [n + 3 for n in lens]


def test_out_of_range_token_id_raises_index_error(tmp_path):
    """nn.Embedding raises IndexError on an id >= vocab (reference transformer.py:193).  Host-resident ids raise before the
    launch; ids that only exist on the device are flagged by the embedding kernel and raised by generate() at its final
    synchronisation - never a silent clamp."""
    case = Case("dense_bf16")
    w = {k: v.to(BF) for k, v in mo.synth_weights(case.args, seed=case.meta["seed"], dtype=BF).items()}
    m = _load(tmp_path, case.args, w)
    V = case.args.vocab_size
    with pytest.raises(IndexError):
        m.forward(torch.tensor([1, V, 3]), [3])                      # CPU tensor: checked on the host
    from mistral_inference.generate import generate
    with pytest.raises(IndexError):
        generate([[1, 5, V + 7, 9]], m, max_tokens=2, temperature=0.0)  # prompts are host data: raised before any launch
    # ids that exist only on the device: the embedding kernel flags them, the next health check raises
    m.forward(torch.tensor([1, V + 7, 3], device="cuda"), [3])
    with pytest.raises(IndexError):
        m._backend.raise_if_flagged()
    m._backend.raise_if_flagged()                                       # the flag was consumed
    toks, _ = generate([[1, 5, 9]], m, max_tokens=2, temperature=0.0)   # ... and the model is usable again
    assert len(toks[0]) == 2
