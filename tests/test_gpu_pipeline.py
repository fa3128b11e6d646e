"""The HIP pipeline path with 2 ranks: real kernels, per-rank layer ranges, device-tensor send/recv and the logits
broadcast.  Both ranks share the one GPU of the test box, so the transport is gloo (RCCL refuses two ranks on one
device); on a multi-GPU node the same code runs over "nccl" = RCCL (bench.py --gpus N)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, tmp, q):
    for p in (os.path.join(ROOT, "mistral-inference_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import Case
        from hip_util import write_checkpoint
        from mistral_inference.generate import generate
        from mistral_inference.transformer import Transformer
        case = Case(name)
        folder = os.path.join(tmp, "ckpt")
        if rank == 0:
            write_checkpoint(folder, case.args, case.weights())
        dist.barrier()
        m = Transformer.from_folder(folder, max_batch_size=case.max_batch_size, num_pipeline_ranks=world, device="cuda:0",
                                    dtype=torch.bfloat16)
        prompts = case.prompts if rank == 0 else [[0] * len(p) for p in case.prompts]
        toks, lps = generate(prompts, m, max_tokens=case.max_tokens, temperature=0.0, chunk_size=case.chunk_size)
        q.put((rank, toks, lps, m.n_local_layers, sorted(m.layers.keys())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["dense_bf16", "swa_chunk_bf16", "moe_bf16"])
def test_two_stage_pipeline_on_hip(name, tmp_path):
    from golden_util import Case
    case = Case(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 500) + sum(map(ord, name)) % 97  # a fresh port per case (no TIME_WAIT reuse)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_toks, ref_lps = case.tokens(), case.logprobs()
    (_, t0, lp0, n0, k0), (_, t1, lp1, n1, k1) = res
    assert (n0, n1) == (1, 1) and k0 == ["0"] and k1 == ["1"]
    assert t0 == t1  # every rank samples from the same broadcast logits
    for b, (mine, ref) in enumerate(zip(t0, ref_toks)):
        n = next((i for i, (x, y) in enumerate(zip(mine, ref)) if x != y), len(ref))
        assert n >= 1, (name, b, mine, ref)
        npl = len(case.prompts[b]) - 1 + n
        assert max(abs(x - y) for x, y in zip(lp0[b][:npl], ref_lps[b][:npl])) <= 6e-2


def test_bench_two_ranks_prints_one_json_line():
    """bench.py's N > 1 path (pipeline stages, max-over-ranks timing, rank-0 report, orderly shutdown) with two ranks
    sharing this box's GPU over gloo; on a multi-GPU node the same code runs over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29377", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--layers", "4", "--prefill", "256", "--mixtral-layers", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and "cpu_baseline" not in d
    # north_star's multi-GPU model: a Mixtral sub-measurement over the same stages (8x7B dims for N < 8, layer-truncated here)
    mx = d["mixtral"]
    assert "Mixtral-8x7B" in mx["model"] and mx["tokens_per_s"] > 0 and 0 < mx["hbm_roofline_frac"] < 1 and mx["prefill_tokens_per_s"] > 0


def test_plain_bench_invocation_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment (the shape of the driver's N = 1 command with another N) must
    start its two ranks itself and still print exactly one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MI_DIST_BACKEND"] = "gloo"  # two ranks on this box's one GPU (RCCL refuses that); "nccl" on a multi-GPU node
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--layers", "2",
           "--prefill", "128", "--no-mixtral"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and "pp2" in d["config"]["parallelism"]
