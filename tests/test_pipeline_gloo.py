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


def _worker(rank, world, port, name, q, session=False):
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
        m = cls(a, pipeline_rank=rank, num_pipeline_ranks=world, backend=OracleStackBackend())
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
        q.put((rank, toks, lps, m.n_local_layers) + ((traffic,) if session else ()))
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
