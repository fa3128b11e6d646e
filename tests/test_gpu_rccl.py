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


def test_interleaved_decoder_tick_replays_from_a_graph_with_the_exchange_inside(comm):
    """ONE tick of the pipeline's throughput mode (`pipeline_decode.InterleavedDecoder`: the stage call
    + the grouped RCCL exchange) as a hipGraph under the C-ABI transport.  A 1-GPU box cannot host two ranks, so the stage here is
    first and last at once and its exchange is a grouped send + recv to itself through `RcclComm` - the same ncclSend / ncclRecv
    nodes a real stage's tick holds.  The replayed ticks must give the tokens and log-probabilities of a plain GreedySession."""
    import mistral_oracle as mo
    from test_gpu_engine import SHAPES, _model
    from mistral_inference.cache import BufferCache
    from mistral_inference.pipeline_decode import InterleavedDecoder
    m, _ = _model(mo.OracleArgs(**SHAPES["gqa4_window_wraps"]), seed=31)
    a = m.args
    ids = torch.randint(0, a.vocab_size, (40,), generator=torch.Generator().manual_seed(5)).cuda()

    def fresh():
        c = BufferCache(m.n_local_layers, 1, 128, a.n_kv_heads, a.head_dim, a.sliding_window, device="cuda", dtype=BF)
        c.reset()
        first = torch.argmax(m.forward(ids, [40], c)[-1:], dim=-1)
        return c, first

    c0, first = fresh()
    sess = m.greedy_session(c0, first, graph=False)
    sess.run(24)
    want_t, want_l = sess.collect()
    del sess

    sink = torch.zeros(1, dtype=torch.long, device="cuda")

    class SelfExchange(InterleavedDecoder):
        def _exchange(self, send_t, dst, recv_t, src):   # the hop of a real stage, to this very rank
            comm.exchange_with_self(send_t, sink)

    c1, first1 = fresh()
    assert torch.equal(first, first1)
    dec = SelfExchange(m, [c1], first1, graph=True)
    assert dec._use_graph
    t1, l1 = dec.run(10)
    t2, l2 = dec.run(14)
    torch.cuda.synchronize()
    assert dec.ticks_replayed >= 22                       # everything but the first (eager) tick and the capturing one... is a replay
    got_t, got_l = torch.cat([t1, t2]), torch.cat([l1, l2])
    assert torch.equal(got_t, want_t) and torch.allclose(got_l, want_l, atol=2e-5)
    assert int(sink[0]) == int(got_t[-1, 0])              # the last tick's sample really travelled through the RCCL nodes
    assert dec.tick_host_us > 0
