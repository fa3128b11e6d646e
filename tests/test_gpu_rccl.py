"""The RCCL transport of the pipeline exchange steps (`mi_rccl_*`, mistral_inference/distributed.py) on ONE GPU: a
world-size-1 communicator exchanging with itself exercises library resolution, communicator set-up, stream-ordered
ncclSend / ncclRecv / ncclBroadcast through the C ABI and their capture in a hipGraph (what a pipeline rank's decode
step replays).  Ordering across real ranks is covered over gloo (tests/test_pipeline_gloo.py, test_gpu_pipeline.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def comm():
    from mistral_inference.distributed import RcclComm
    c = RcclComm(1, 0, RcclComm.unique_id())
    yield c
    c.close()


def test_self_exchange_and_broadcast(comm):
    src = torch.randn(5, 4096, device="cuda").to(BF)
    dst = torch.zeros_like(src)
    comm.exchange_with_self(src, dst)
    lg = torch.randn(3, 32768, device="cuda")
    keep = lg.clone()
    comm.broadcast(lg, src=0)
    torch.cuda.synchronize()
    assert torch.equal(src, dst) and torch.equal(lg, keep)


def test_transfers_are_stream_ordered_and_graph_capturable(comm):
    """Producer kernel -> send/recv -> consumer kernel on one stream without a host sync, then the same sequence replayed
    from a hipGraph with new data: the transfers behave like any other node of the decode step's graph."""
    x = torch.randn(1, 4096, device="cuda").to(BF)
    a, b = torch.empty_like(x), torch.empty_like(x)
    out = torch.empty_like(x)

    def step():
        a.copy_(x * 2)                 # "previous stage" produces
        comm.exchange_with_self(a, b)  # activations travel
        comm.broadcast(b, src=0)       # (logits broadcast of the last stage)
        out.copy_(b + 1)               # "next stage" consumes

    step()
    torch.cuda.synchronize()
    assert torch.equal(out, (x * 2).to(BF) + 1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for seed in (1, 2):
        x.copy_(torch.randn(1, 4096, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)).to(BF))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, (x * 2).to(BF) + 1), seed


def test_pipeline_comm_selection_without_process_group():
    """No process group (single rank): pipeline_comm falls back to the torch.distributed transport object; a model
    built for one rank never touches either."""
    from mistral_inference.distributed import TorchDistComm, pipeline_comm
    assert isinstance(pipeline_comm(torch.device("cuda")), TorchDistComm)
