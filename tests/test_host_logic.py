"""Host-side logic that needs no GPU: params.json decoding, ring sizing / metadata, rank-filtered loading."""
import pytest
import torch

import mistral_oracle as mo
from golden_util import Case
from oracle_backend import OracleStackBackend


def test_args_from_dict_aliases_and_unknown_keys():
    from mistral_inference.args import TransformerArgs
    p = dict(dim=8, n_layers=2, head_dim=128, hidden_dim=16, n_heads=2, n_kv_heads=1, norm_eps=1e-5, vocab_size=10,
             _sliding_window=7, moe=dict(num_experts=8, num_experts_per_tok=2), bogus_key=1)
    a = TransformerArgs.from_dict(p)
    assert a.sliding_window == 7 and a.moe.num_experts == 8 and a.rope_theta is None and a.max_batch_size == 0
    with pytest.raises(AssertionError):
        TransformerArgs.from_dict(dict(p, sliding_window=3))  # both spellings given (reference args.py:56)
    with pytest.raises(AssertionError):
        TransformerArgs.from_dict(dict(p, model_type="mamba"))


def test_cache_sizes():
    from mistral_inference.cache import RotatingBufferCache, BufferCache, get_cache_sizes
    assert RotatingBufferCache is BufferCache
    assert get_cache_sizes(4, 100, None) == [100] * 4
    assert get_cache_sizes(4, 100, 4096) == [4096] * 4          # an int window larger than the run still allocates W
    assert get_cache_sizes(4, 100, [8, None]) == [8, 100, 8, 100]
    with pytest.raises(AssertionError):
        get_cache_sizes(3, 100, [8, None])


def test_metadata_worked_example():
    """W=3, seqlens=[5,7,2], kv_seqlens=[1,1,3]."""
    from mistral_inference.cache import BufferCache
    from mistral_inference import _hip
    c = BufferCache(1, 3, 16, 2, 128, sliding_window=3)
    c.init_kvseqlens(3)
    c.update_seqlens([1, 1, 3])
    (md,) = c.get_input_metadata([5, 7, 2])
    assert md.to_cache_mask.int().tolist() == [0, 0, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 1]
    assert md.positions.tolist() == [1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 6, 7, 3, 4]
    assert md.cache_positions.tolist() == [0, 1, 2, 5, 3, 4, 6, 7]
    assert md.cached_elements.tolist() == [3, 3, 2]
    assert md.prefill and md.batch.branch == _hip.BRANCH_PREFILL
    assert md.batch.q_start.tolist() == [0, 5, 12, 14] and md.batch.kv_before.tolist() == [1, 1, 3]
    assert md.batch.tok_seq.tolist() == [0] * 5 + [1] * 7 + [2] * 2
    c.update_seqlens([5, 7, 2])
    assert c.kv_seqlens.tolist() == [6, 8, 5]
    b = c.batch_metadata([1, 1, 1])
    assert b.branch == _hip.BRANCH_DECODE
    with pytest.raises(AssertionError, match="did you forget to reset cache"):
        c.batch_metadata([1, 1])
    c.reset()
    assert c.batch_metadata([1, 1]).branch == _hip.BRANCH_PREFILL  # first call after reset is a (first) prefill


def _model(case, rank, world):
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    a = TransformerArgs.from_dict(case.params)
    a.max_batch_size = case.max_batch_size
    return Transformer(a, pipeline_rank=rank, num_pipeline_ranks=world, backend=OracleStackBackend())


def test_rank_filtered_load_and_unexpected_key():
    case = Case("dense_fp32")
    w = case.weights()
    m0, m1 = _model(case, 0, 2), _model(case, 1, 2)
    assert list(m0.layers.keys()) == ["0"] and list(m1.layers.keys()) == ["1"]   # global ids (transformer.py:97)
    assert m0.tok_embeddings is not None and m0.norm is None and m0.output is None
    assert m1.tok_embeddings is None and m1.norm is not None and m1.output is not None
    m0.load_state_dict(w, assign=True)
    m1.load_state_dict(w, assign=True)
    assert torch.equal(m1.layers["1"].attention.wq.weight, w["layers.1.attention.wq.weight"])
    with pytest.raises(ValueError, match="Unexpected key"):
        m0.load_state_dict(dict(w, stray=torch.zeros(1)))
    three = _model(Case("swa_list_fp32"), 1, 3)  # ceil(2/3) = 1 layer per rank, rank 2 gets none
    assert list(three.layers.keys()) == ["1"]


@pytest.mark.parametrize("name", ["dense_fp32", "swa_chunk_fp32", "moe_fp32"])
def test_generate_host_loop_matches_reference(name):
    """generate() bookkeeping (chunking, logprob order, return shapes) with the oracle standing in for the
    kernels, against the unmodified reference's outputs."""
    from mistral_inference.generate import generate
    case = Case(name)
    m = _model(case, 0, 1)
    m.load_state_dict(case.weights(), assign=True)
    toks, lps = generate(case.prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
    assert toks == case.tokens()
    for a, b in zip(lps, case.logprobs()):
        assert len(a) == len(b) and max(abs(x - y) for x, y in zip(a, b)) < 2e-5
    gen, lps0 = generate(case.prompts, m, max_tokens=0, temperature=0.0, chunk_size=case.chunk_size)
    assert gen == [] and [len(x) for x in lps0] == [len(p) - 1 for p in case.prompts]


@pytest.mark.parametrize("name", ["dense_fp32", "swa_chunk_fp32"])
def test_generate_fused_prompt_logprob_bookkeeping(name):
    """generate()'s other prompt route - per-row target ids in, (logprob per row, last-row logits) out - with a CPU
    stand-in for `prompt_logprobs`: targets, ignored rows, chunk seams and the logprob order must reproduce the
    reference's outputs exactly like the [T, V] route does."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.generate import generate
    from mistral_inference.transformer import Transformer

    class CpuFused(Transformer):
        prompt_logprobs_any_device = True
        calls = 0

        def prompt_logprobs(self, input_ids, seqlens, cache, targets, images=None):
            CpuFused.calls += 1
            logits = self.forward(input_ids, seqlens, cache)
            lsm = torch.log_softmax(logits, dim=-1)
            t = targets.long()
            assert t.shape == (input_ids.numel(),) and int((t < 0).sum()) == len(seqlens)  # last row of every sequence
            lp = lsm.gather(1, t.clamp(min=0)[:, None])[:, 0]
            ends = torch.tensor(seqlens).cumsum(0) - 1
            assert bool((t[ends] < 0).all())
            return lp, logits.index_select(0, ends)

    case = Case(name)
    a = TransformerArgs.from_dict(case.params)
    a.max_batch_size = case.max_batch_size
    m = CpuFused(a, backend=OracleStackBackend())
    m.load_state_dict(case.weights(), assign=True)
    toks, lps = generate(case.prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
    assert CpuFused.calls >= 1 and toks == case.tokens()
    for x, y in zip(lps, case.logprobs()):
        assert len(x) == len(y) and max(abs(p - q) for p, q in zip(x, y)) < 2e-5


def test_generate_eos_and_sampling():
    from mistral_inference.generate import generate, sample_top_p
    case = Case("dense_fp32")
    m = _model(case, 0, 1)
    m.load_state_dict(case.weights(), assign=True)
    ref = case.tokens()
    eos = ref[0][2]
    toks, _ = generate(case.prompts[:1], m, max_tokens=6, temperature=0.0, eos_id=eos)
    assert toks == [ref[0][:2]]  # stops when every sequence has produced eos; eos itself is not emitted
    torch.manual_seed(0)
    probs = torch.tensor([[0.5, 0.25, 0.2, 0.05]])
    picks = {int(sample_top_p(probs, 0.8)) for _ in range(200)}
    assert picks == {0, 1, 2}  # mass before token 3 is 0.95 > 0.8 -> never drawn; token 2 (0.75 before) is kept
    torch.manual_seed(1)
    toks, lps = generate(case.prompts, m, max_tokens=3, temperature=0.7)
    assert all(len(t) == 3 for t in toks) and all(x <= 0 for l in lps for x in l)


@pytest.mark.parametrize("name", ["dense_fp32", "swa_chunk_fp32"])
def test_generate_fused_greedy_bookkeeping(name):
    """generate()'s temperature-0 route through a greedy session (first sample from the prompt's last logits, then chunks of
    session steps read back at once, EOS cut inside a chunk) with a CPU stand-in for `GreedySession`: tokens and logprobs
    must equal the reference's, with and without an eos_id, for every cut position."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.generate import generate
    from mistral_inference.transformer import Transformer

    class CpuSession:
        HIST = 1024

        def __init__(self, model, cache, first):
            self.m, self.cache, self.tok, self.out = model, cache, first.clone(), []

        def run(self, n):
            for _ in range(n):
                logits = self.m.forward(self.tok, [1] * self.tok.numel(), self.cache)
                self.tok = torch.argmax(logits, dim=-1)
                lp = torch.log_softmax(logits, dim=-1).gather(1, self.tok[:, None])[:, 0]
                self.out.append((self.tok.clone(), lp))

        def collect(self, n):
            got, self.out = self.out[:n], self.out[n:]
            assert len(got) == n
            return torch.stack([t for t, _ in got]), torch.stack([l for _, l in got])

    class CpuGreedy(Transformer):
        greedy_session_any_device = True
        sessions = 0

        def greedy_session(self, cache, first_tokens, graph=True):
            CpuGreedy.sessions += 1
            return CpuSession(self, cache, first_tokens)

    case = Case(name)
    a = TransformerArgs.from_dict(case.params)
    a.max_batch_size = case.max_batch_size
    m = CpuGreedy(a, backend=OracleStackBackend())
    m.load_state_dict(case.weights(), assign=True)
    toks, lps = generate(case.prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
    assert CpuGreedy.sessions >= 1 and toks == case.tokens()
    for x, y in zip(lps, case.logprobs()):
        assert len(x) == len(y) and max(abs(p - q) for p, q in zip(x, y)) < 2e-5
    # EOS: the reference stops BEFORE the step at which every sequence has produced eos (generate.py:128-132); with one
    # sequence that is the position of the first eos, whichever token of the run is declared eos
    ref = case.tokens()[0]
    for cut in range(len(ref)):
        eos = ref[cut]
        first = ref.index(eos)
        t1, l1 = generate(case.prompts[:1], m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size, eos_id=eos)
        assert t1 == ([ref[:first]] if first else []), (cut, t1)
        assert len(l1[0]) == len(case.prompts[0]) - 1 + first
    # several sequences: generation continues until ALL have hit eos - same tokens and logprobs as the host loop
    seen = sorted({t for r in case.tokens() for t in r})
    for eos in seen[:6]:
        m.fused_greedy = True
        tf, lf = generate(case.prompts, m, max_tokens=case.max_tokens + 2, temperature=0.0, chunk_size=case.chunk_size, eos_id=eos)
        m.fused_greedy = False
        th, lh = generate(case.prompts, m, max_tokens=case.max_tokens + 2, temperature=0.0, chunk_size=case.chunk_size, eos_id=eos)
        assert tf == th, (eos, tf, th)
        assert all(len(x) == len(y) and max([abs(p - q) for p, q in zip(x, y)] + [0]) < 2e-5 for x, y in zip(lf, lh))
    m.fused_greedy = True
    # one token asked for: the session is created but never stepped
    t0, _ = generate(case.prompts, m, max_tokens=1, temperature=0.0, chunk_size=case.chunk_size)
    assert t0 == [r[:1] for r in case.tokens()]


def test_interleave_kv_and_unrotate_semantics():
    """CacheView.interleave_kv (reference cache.py:94-117): per sequence its cached tokens in position order (ring
    unrotated, at most W of them) followed by its new tokens; rings that have not wrapped, wrapped exactly, and wrapped
    twice."""
    from mistral_inference.cache import CacheInputMetadata, CacheView, unrotate
    B, W, H, D = 4, 4, 2, 8
    g = torch.Generator().manual_seed(0)
    hist = [torch.randn(n, H, D, generator=g) for n in (0, 3, 8, 10)]   # everything each sequence has seen so far
    ck, cv = torch.zeros(B, W, H, D), torch.zeros(B, W, H, D)
    for b, hb in enumerate(hist):
        for p in range(hb.shape[0]):
            ck[b, p % W] = hb[p]
            cv[b, p % W] = -hb[p]
    new = [2, 1, 3, 2]
    xk = torch.randn(sum(new), H, D, generator=g)
    md = CacheInputMetadata(positions=None, to_cache_mask=None, cached_elements=None, cache_positions=None, prefill=True,
                            mask=None, seqlens=new)
    k, v = CacheView(ck, cv, md, torch.tensor([h.shape[0] for h in hist])).interleave_kv(xk, -xk)
    want, o = [], 0
    for hb, n in zip(hist, new):
        want += [hb[max(0, hb.shape[0] - W):], xk[o:o + n]]
        o += n
    assert torch.equal(k, torch.cat(want)) and torch.equal(v, -torch.cat(want))
    assert torch.equal(unrotate(ck[3], 10), hist[3][6:]) and torch.equal(unrotate(ck[1], 3), hist[1])
    fresh = CacheView(ck, cv, md, torch.zeros(B, dtype=torch.long))
    assert fresh.interleave_kv(xk, xk)[0] is xk   # nothing cached: the inputs come back (cache.py:101-103)


def test_out_of_range_token_ids_raise_like_nn_embedding():
    """Host-resident ids are validated before any launch (reference: nn.Embedding raises IndexError, transformer.py:193);
    device-resident ids are flagged by the kernel instead (tests/test_gpu_model.py)."""
    import pytest as _pytest
    import torch as _torch
    from mistral_inference import _hip
    _hip.check_ids_on_host(_torch.tensor([0, 5, 511]), 512)
    _hip.check_ids_on_host(_torch.tensor([], dtype=_torch.long), 512)
    for bad in ([512], [3, -1]):
        with _pytest.raises(IndexError):
            _hip.check_ids_on_host(_torch.tensor(bad), 512)


def test_pipeline_transport_selection_without_a_gpu_process_group():
    """No "nccl" group on a GPU -> the reference's own torch.distributed transport object (what the gloo tests drive)."""
    import torch as _torch
    from mistral_inference.distributed import TorchDistComm, pipeline_comm
    assert isinstance(pipeline_comm(_torch.device("cpu")), TorchDistComm)


def test_buffer_cache_head_major_storage_keeps_the_reference_shape(monkeypatch):
    """BufferCache stores its rings head-major ([max_batch, n_kv_heads, W, head_dim]: one kv head's slots contiguous, DESIGN.md
    section 2) and shows them in the reference's shape (cache.py:163-167): indexing, CacheView.key / .value, interleave_kv,
    .to() behave as with the reference's layout; MI_KV_LAYOUT=0 allocates that layout; `_hip.kv_layout_of` tells them apart and
    rejects anything else."""
    from mistral_inference import _hip
    from mistral_inference.cache import BufferCache
    c = BufferCache(2, 3, 10, 4, 8, sliding_window=[6, None], dtype=torch.float32)
    assert [tuple(c.cache_k[i].shape) for i in range(2)] == [(3, 6, 4, 8), (3, 10, 4, 8)]
    assert c.kv_layout == _hip.KV_HEAD_MAJOR and not c.cache_k[0].is_contiguous()
    assert c.cache_k[1].stride() == (4 * 10 * 8, 8, 10 * 8, 1)            # slot stride = head_dim, head stride = W * head_dim
    c.cache_k[0][1, 5, 2] = torch.arange(8.0)
    assert torch.equal(c.cache_k[0].permute(0, 2, 1, 3)[1, 2, 5], torch.arange(8.0))   # the same element through the storage's own shape
    c.init_kvseqlens(2)
    view = c.get_view(0, None)
    assert tuple(view.key.shape) == (2, 6, 4, 8) and view.max_seq_len == 6
    c2 = c.to("cpu", torch.bfloat16)
    assert c2.kv_layout == _hip.KV_HEAD_MAJOR and c2.cache_k[0].dtype == torch.bfloat16 and float(c2.cache_k[0][1, 5, 2, 3]) == 3.0
    monkeypatch.setenv("MI_KV_LAYOUT", "0")
    r = BufferCache(2, 3, 10, 4, 8, sliding_window=[6, None], dtype=torch.float32)
    assert r.kv_layout == _hip.KV_SLOT_MAJOR and r.cache_k[0].is_contiguous() and tuple(r.cache_k[0].shape) == (3, 6, 4, 8)
    with pytest.raises(ValueError):
        _hip.kv_layout_of(torch.zeros(3, 6, 4, 16)[..., ::2])
    with pytest.raises(AssertionError):
        r.cache_k[1] = c.cache_k[1]
        _ = r.kv_layout
    # shapes whose two layouts coincide (one kv head, or one slot) count as the reference's
    assert _hip.kv_layout_of(torch.zeros(2, 1, 5, 8).permute(0, 2, 1, 3)) == _hip.KV_SLOT_MAJOR


def test_sample_never_communicates_and_takes_an_agreed_seed():
    """`sample()` must not issue a collective (the reference's has none, generate.py:151-159) - ranks of a
    data-parallel job, or ranks that call it a different number of times, would hang or couple their seeds.  The pipeline's seed
    agreement lives in generate(), over the model's own communicator; sample() only accepts the agreed (seed, offset)."""
    import inspect
    from mistral_inference import generate as G
    src = inspect.getsource(G.sample)
    assert "torch.distributed" not in src and ".broadcast(" not in src and "all_reduce" not in src and "dist." not in src
    assert {"seed", "offset"} <= set(inspect.signature(G.sample).parameters)
    gsrc = inspect.getsource(G.generate)
    assert "model.pp_comm.broadcast" in gsrc and "num_pipeline_ranks" in gsrc
    # greedy and the torch path keep the reference's behaviour on CPU
    logits = torch.tensor([[0.1, 2.0, -1.0], [3.0, 0.0, 0.5]])
    assert G.sample(logits, temperature=0.0, top_p=0.8).tolist() == [1, 0]
    torch.manual_seed(0)
    a = G.sample(logits, temperature=0.7, top_p=0.8)
    torch.manual_seed(0)
    assert torch.equal(a, G.sample(logits, temperature=0.7, top_p=0.8, seed=123, offset=4))  # (CPU path: torch's own generator)
