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


def _worker(rank, world, port, name, q):
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
        m = Transformer(a, pipeline_rank=rank, num_pipeline_ranks=world, backend=OracleStackBackend())
        m.load_state_dict(case.weights(), assign=True)
        prompts = case.prompts if rank == 0 else [[0] * len(p) for p in case.prompts]  # reference main.py:169-170
        toks, lps = generate(prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
        q.put((rank, toks, lps, m.n_local_layers))
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
