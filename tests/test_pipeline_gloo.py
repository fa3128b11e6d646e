"""Pipeline-parallel host path on CPU: world_size 2 over gloo, the oracle standing in for the kernels.

Covers what bench.py --gpus N / the RCCL path rely on but a 1-GPU box cannot show: contiguous layer ranges,
rank-filtered loading, recv -> stack -> send order, the logits broadcast from the last rank, per-rank caches."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, q, session=False, sabotage=None):
    for p in (os.path.join(ROOT, "mistral-inference_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import Case
        from mistral_inference.args import TransformerArgs
        from mistral_inference.generate import generate
        from mistral_inference.transformer import Transformer
        from oracle_backend import OracleStackBackend
        case = Case(name)
        a = TransformerArgs.from_dict(case.params)
        a.max_batch_size = case.max_batch_size
        cls = Transformer
        traffic = []
        if session:
            # the product's GreedySession on every stage (CPU tensors, the oracle as the stack): the per-token protocol of the
            # pipeline - activations forward, the 8-byte sample back to stage 0, history broadcast once per collect()
            class SessionOnCpu(Transformer):
                greedy_session_any_device = True
            cls = SessionOnCpu
        # sabotage = (stage, decode step): that stage's "engine" fails its residency gate from that step on (writes nothing, 0x700)
        fail = sabotage[1] if sabotage is not None and sabotage[0] == rank else None
        m = cls(a, pipeline_rank=rank, num_pipeline_ranks=world, backend=OracleStackBackend(fail_from_step=fail))
        m.load_state_dict(case.weights(), assign=True)
        if session:
            comm = m.pp_comm

            class Recorder:
                def send(self, t, dst):
                    traffic.append(("send", tuple(t.shape), t.numel() * t.element_size()))
                    comm.send(t, dst)

                def recv(self, t, src):
                    comm.recv(t, src)

                def broadcast(self, t, src):
                    traffic.append(("bcast", tuple(t.shape), t.numel() * t.element_size()))
                    comm.broadcast(t, src)
            m._pp_comm = Recorder()
        prompts = case.prompts if rank == 0 else [[0] * len(p) for p in case.prompts]  # reference main.py:169-170
        toks, lps = generate(prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
        q.put((rank, toks, lps, m.n_local_layers) + ((traffic,) if session else ()) + ((m._backend.status, m._backend.steps, m._backend.fail_from_step),) * (sabotage is not None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["dense_fp32", "swa_chunk_fp32"])
def test_two_stage_pipeline_matches_single_process(name):
    from golden_util import Case
    case = Case(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1500) + sum(map(ord, name)) % 97  # a fresh port per case
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, toks, lps, n_local in res:
        assert n_local == 1
        # every rank samples from the same broadcast logits -> identical tokens everywhere (no token broadcast)
        assert toks == case.tokens(), rank
        if rank == 0 or True:  # logprobs of generated tokens agree on all ranks; prompt logprobs need the real ids
            gen = case.max_tokens
            for a, b in zip(lps, case.logprobs()):
                assert max(abs(x - y) for x, y in zip(a[-gen:], b[-gen:])) < 2e-5
        if rank == 0:
            for a, b in zip(lps, case.logprobs()):
                assert max(abs(x - y) for x, y in zip(a, b)) < 2e-5


@pytest.mark.parametrize("name", ["dense_fp32", "swa_chunk_fp32"])
def test_greedy_session_across_two_stages_carries_the_sample_not_the_logits(name):
    """generate() at temperature 0 over two pipeline stages through the product's GreedySession: tokens and logprobs equal
    the single-process reference outputs on BOTH ranks, and at decode nothing of vocabulary size crosses between the ranks -
    per token one [B, dim] activation hop forward and one [B] int64 sample back; the other rank learns the tokens from one
    history broadcast per collect() (the reference: [B, vocab] logits to every rank per token, transformer.py:236-237)."""
    from golden_util import Case
    case = Case(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 1500) + sum(map(ord, name)) % 97
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, V, D = len(case.prompts), case.params["vocab_size"], case.params["dim"]
    for rank, toks, lps, n_local, traffic in res:
        assert toks == case.tokens(), rank
        gen = case.max_tokens
        for a, b in zip(lps, case.logprobs()):
            assert max(abs(x - y) for x, y in zip(a[-gen:], b[-gen:])) < 2e-5
        # decode traffic: no [B, V] broadcast (the prompt's broadcasts have T = sum of the chunk lengths rows, never B)
        assert not [t for t in traffic if t[0] == "bcast" and t[1] == (B, V)], traffic
        sample_hops = [t for t in traffic if t[0] == "send" and t[1] == (B,)]
        act_hops = [t for t in traffic if t[0] == "send" and t[1] == (B, D)]
        if rank == 1:
            assert len(sample_hops) == gen - 1 and all(t[2] == 8 * B for t in sample_hops)   # 8 bytes per sequence per token
        else:
            assert len(act_hops) == gen - 1
        hist = [t for t in traffic if t[0] == "bcast" and len(t[1]) == 2 and t[1][1] == B and t[1][0] <= gen]
        assert len(hist) == 2   # tokens + logprobs, ONE collect for the whole generation (no eos_id)


def _interleaved_worker(rank, world, port, q, n_dec):
    for p in (os.path.join(ROOT, "mistral-inference_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mistral_oracle as mo
        from mistral_inference.args import TransformerArgs
        from mistral_inference.cache import BufferCache
        from mistral_inference.pipeline_decode import InterleavedDecoder
        from mistral_inference.transformer import Transformer
        from oracle_backend import OracleStackBackend
        params, w, prompts = _interleaved_case(world)
        a = TransformerArgs.from_dict(params)
        a.max_batch_size = 1
        m = Transformer(a, pipeline_rank=rank, num_pipeline_ranks=world, backend=OracleStackBackend())
        m.load_state_dict(w, assign=True)
        comm = m.pp_comm
        traffic = []

        class Recorder:
            def send(self, t, dst):
                comm.send(t, dst)

            def recv(self, t, src):
                comm.recv(t, src)

            def broadcast(self, t, src):
                comm.broadcast(t, src)

            def exchange(self, send_t, dst, recv_t, src):
                traffic.append((None if send_t is None else (dst, send_t.numel() * send_t.element_size()),
                                None if recv_t is None else (src, recv_t.numel() * recv_t.element_size())))
                comm.exchange(send_t, dst, recv_t, src)
        m._pp_comm = Recorder()
        caches, first = [], []
        for pr in prompts:  # every sequence prefilled through the ordinary pipeline forward, into its own rings
            c = BufferCache(m.n_local_layers, 1, len(pr) + n_dec + 2, a.n_kv_heads, a.head_dim, a.sliding_window, dtype=torch.float32)
            c.reset()
            ids = torch.tensor(pr if rank == 0 else [0] * len(pr), dtype=torch.long)
            logits = m.forward(ids, [len(pr)], c)
            caches.append(c)
            first.append(int(torch.argmax(logits[-1])))
        dec = InterleavedDecoder(m, caches, torch.tensor(first))
        toks, lps = dec.run(n_dec)
        assert dec.tick_host_us > 0.0  # host time per tick of the loop (bench.py: pipeline_throughput.tick_host_us)
        toks2, _ = dec.run(2)  # a second call continues every sequence
        q.put((rank, first, toks.tolist(), lps.tolist(), toks2.tolist(), traffic))
    finally:
        dist.destroy_process_group()


def _interleaved_case(world):
    import mistral_oracle as mo
    params = dict(dim=64, n_layers=2 * world - 1, head_dim=32, hidden_dim=128, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=97,
                  sliding_window=6)
    args = mo.OracleArgs.from_params(params)
    w = {k: v.float() for k, v in mo.synth_weights(args, seed=11, dtype=torch.bfloat16).items()}
    prompts = [[(5 * i + 3 * j + 1) % 97 for i in range(4 + 3 * j)] for j in range(world)]
    return params, w, prompts


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_interleaved_decoder_one_sequence_per_stage(world):
    """pipeline_decode.InterleavedDecoder over gloo: `world` sequences through `world` stages, every stage busy on a different
    sequence each tick.  Tokens and log-probabilities of EVERY sequence equal decoding it alone in one process (the oracle's
    generate), on every rank; per tick a stage posts exactly one grouped exchange carrying at most one [1, dim] activation
    forward and - on the ring's closing link - 8 bytes of sample; uneven layer split (2 world - 1 layers), a window that wraps."""
    import mistral_oracle as mo
    n_dec = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 1500) + world
    procs = [ctx.Process(target=_interleaved_worker, args=(r, world, port, q, n_dec)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params, w, prompts = _interleaved_case(world)
    args = mo.OracleArgs.from_params(params)
    ref_t, ref_lp = [], []
    for pr in prompts:
        t, lp = mo.generate([pr], mo.OracleModel(args, w), max_tokens=n_dec + 3, max_batch_size=1)
        ref_t.append(t[0])
        ref_lp.append(lp[0][len(pr) - 1:])
    D = params["dim"]
    for rank, first, toks, lps, toks2, traffic in res:
        for j in range(world):
            assert first[j] == ref_t[j][0], (rank, j)
            assert [row[j] for row in toks] == ref_t[j][1:n_dec + 1], (rank, j)
            assert [row[j] for row in toks2] == ref_t[j][n_dec + 1:n_dec + 3], (rank, j)
            assert max(abs(row[j] - x) for row, x in zip(lps, ref_lp[j][1:n_dec + 1])) < 2e-5
        # one exchange per tick; what leaves: activations (4 * D bytes, fp32 here) or, from the last stage, the 8-byte sample
        assert len(traffic) == (n_dec + 2) * world + 2 * (world - 1)
        sends = [s for s, _ in traffic if s is not None]
        assert len(sends) == (n_dec + 2) * world
        assert all(b == (8 if rank == world - 1 else 4 * D) and d == (rank + 1) % world for d, b in sends)


@pytest.mark.parametrize("stage,step", [(1, 2), (0, 0), (0, 3)])
def test_pipeline_rolls_back_in_lock_step_after_a_residency_failure(stage, step):
    """A decode step whose engine launch fails its residency gate on ONE pipeline stage (status 0x700: nothing written from that
    step on, the stage keeps forwarding stale activations) used to cost the whole generation under pipeline parallelism (round-5
    DESIGN section 7 "known limitation").  Now collect() learns over the bootstrap group how many steps completed on EVERY stage,
    all stages rewind to that step - positions, step counter, stage 0's input id - and re-run the rest on the launch path:
    tokens and log-probabilities equal the single-process reference outputs on both ranks, whichever stage failed and whether
    it failed at the session's very first step or later."""
    from golden_util import Case
    name = "dense_fp32"
    case = Case(name)
    assert case.max_tokens - 1 > step + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 1500) + 7 * stage + step
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q, True, (stage, step))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, toks, lps, n_local, traffic, (status, steps, still_armed) in res:
        assert still_armed is None               # the failure really happened and the rollback disarmed it (session_disable_engine)
        assert toks == case.tokens(), (rank, toks, case.tokens())
        gen = case.max_tokens
        for a, b in zip(lps, case.logprobs()):
            assert max(abs(x - y) for x, y in zip(a[-gen:], b[-gen:])) < 2e-5
        assert status == 0                       # cleared by the rollback; the stage finished on the "launch path"
        assert steps == case.max_tokens - 1      # the step counter ends where an undisturbed generation's would
