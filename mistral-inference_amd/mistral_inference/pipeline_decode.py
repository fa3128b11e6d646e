"""Decode throughput of a pipeline: N independent sequences through N stages, one sequence per stage at any time.

The reference's pipeline (transformer.py:195-196,213-214,236-237) is a relay: for a single sequence stage r waits for stage
r - 1, so N GPUs decode no faster than one ("scaling only appears ... with multiple concurrent sequences
(not in the reference)").  This is that mode.  Every stage owns 1 / N of the layers, so a token costs it 1 / N of a step; with
one sequence per stage in flight, every tick every stage runs ONE batch-1 decode call (for Mistral / Mixtral shapes: one launch
of the persistent engine) on a different sequence, and the N results move one stage along a ring:

    tick t, stage r:   sequence (t - r) mod N;   activations -> stage r + 1;   the last stage's sample (8 bytes) -> stage 0

The send and the receive of a stage are ONE grouped exchange per tick (`exchange`: ncclGroupStart / End, or
`batch_isend_irecv`), so the ring cannot deadlock on rendezvous and every link carries one message per tick.  Each call is
still batch = 1, seq = 1 against that sequence's own K/V rings - the arithmetic of a sequence does not depend on the others,
which is what the tests check (tokens identical to decoding every sequence alone).  Aggregate rate = one token per tick =
N / (T_step + N x hop) against 1 / (T_step + N x hop) for the relay.
"""
from __future__ import annotations

import time
from typing import List, Tuple

import torch

from . import _hip
from .cache import BufferCache
from .transformer import GreedyBuffers, Transformer


class InterleavedDecoder:
    """`caches[j]` = this stage's K/V rings of sequence j, already prefilled (every stage ran `model.forward(prompt_j,
    [len(prompt_j)], caches[j])`); `first_tokens[j]` = the token that follows prompt j (used on stage 0).  `run(n)` decodes
    n greedy tokens for every sequence and returns (tokens int64 [n, N], logprobs fp32 [n, N]) on every stage.

    One tick of a stage = ONE native call (its layer range on sequence j; on the last stage it ends in the fused sample, which
    lands in the id buffer of sequence j and in a history ring shared by all sequences, row = the workspace's step counter) +
    ONE grouped exchange.  With the C-ABI transport (`distributed.RcclComm`: stream-ordered ncclSend / ncclRecv) both are
    capturable: steady-state ticks are replayed from a hipGraph per (sequence, activation buffer) - one graph launch per tick,
    no per-tick ctypes / process-group work on the host (`graph=True`, round 6; validated on one GPU with a self-exchange,
    tests/test_gpu_rccl.py; the torch.distributed transport steps eagerly as before)."""

    HIST = 1024

    def __init__(self, model: Transformer, caches: List[BufferCache], first_tokens: torch.Tensor, graph: bool = True):
        self.model = model
        self.rank, self.world = model.pipeline_rank, model.num_pipeline_ranks
        self.n_seq = len(caches)
        assert self.n_seq == self.world, "one sequence in flight per pipeline stage"
        assert first_tokens.numel() == self.n_seq
        for c in caches:
            assert c._seen is not None and len(c._seen) == 1 and c._seen[0] > 0, "prefill every sequence's cache first (batch 1 each)"
        self.caches = caches
        dev = model.device
        self.is_first, self.is_last = self.rank == 0, self.rank == self.world - 1
        # stage 0: the next input id of every sequence; last stage: where the fused sample lands
        self.tok = [first_tokens.reshape(-1)[j:j + 1].to(device=dev, dtype=torch.long).clone() for j in range(self.n_seq)]
        # ONE history ring for all sequences: the last stage's samples land in tick order (row = step counter of its workspace)
        self.hist_tok = torch.zeros((self.HIST, 1), dtype=torch.long, device=dev)
        self.hist_lp = torch.zeros((self.HIST, 1), dtype=torch.float32, device=dev)
        self.bufs = [GreedyBuffers(tok=self.tok[j], lp=torch.zeros(1, dtype=torch.float32, device=dev),
                                   hist_tok=self.hist_tok, hist_lp=self.hist_lp) for j in range(self.n_seq)]
        self.logits = torch.empty((1, model.vocab_size), dtype=torch.float32, device=dev) if self.is_last else None
        self.h = [torch.empty((1, model.args.dim), dtype=model.dtype, device=dev) for _ in range(2)]  # in flight / being filled
        self.steps_done = 0
        self.tick_host_us = 0.0  # host time per tick of the last run() (enqueue cost of one stage call + one grouped exchange)
        from .distributed import RcclComm
        self._use_graph = bool(graph and dev.type == "cuda" and (self.world == 1 or isinstance(model.pp_comm, RcclComm)))
        self._graphs: dict = {}   # (sequence, activation buffer, receive target) -> captured tick
        self._warm = [False] * self.n_seq  # a sequence's first tick on this stage runs eagerly (workspace sizing, engine census)
        self.ticks_replayed = 0

    # -- the device work of one tick: the stage call on sequence j, then the grouped exchange
    def _exchange(self, send_t, dst: int, recv_t, src: int) -> None:
        if self.world > 1:
            self.model.pp_comm.exchange(send_t, dst, recv_t, src)

    def _device_tick(self, j: int, cur: int, recv_t, nxt: int, prv: int) -> None:
        m = self.model
        cache = self.caches[j]
        meta = cache.batch_metadata([1])
        assert meta.branch == _hip.BRANCH_DECODE
        h = self.h[cur]
        m._backend.run_stack(m, h, self.tok[j] if self.is_first else None, meta, cache, self.logits,
                             greedy=self.bufs[j] if self.is_last else None)
        # the sample (8 bytes) goes back to stage 0, the activations [1, dim] on to the next stage
        self._exchange(self.tok[j] if self.is_last else h, nxt, recv_t, prv)

    def _tick(self, j: int, cur: int, recv_t, nxt: int, prv: int) -> None:
        key = (j, cur, None if recv_t is None else recv_t.data_ptr())
        if self._use_graph and self._warm[j]:
            g = self._graphs.get(key)
            if g is None:
                torch.cuda.synchronize(self.model.device)
                g = torch.cuda.CUDAGraph()
                ok = True
                try:
                    with torch.cuda.graph(g):  # (capture enqueues nothing: the tick itself is the replay below)
                        self._device_tick(j, cur, recv_t, nxt, prv)
                except RuntimeError:
                    ok = False
                if self.world > 1 and torch.distributed.is_initialized():  # replay everywhere or nowhere (GreedySession._captured)
                    flag = torch.tensor([1 if ok else 0], device=self.model.device, dtype=torch.int32)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                    ok = bool(int(flag.item()))
                if not ok:
                    self._use_graph, self._graphs = False, {}
                    self._device_tick(j, cur, recv_t, nxt, prv)
                    return
                self._graphs[key] = g
            g.replay()
            self.ticks_replayed += 1
            return
        self._device_tick(j, cur, recv_t, nxt, prv)
        self._warm[j] = True

    def run(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        N = self.world
        if n * N > self.HIST:  # (the shared history ring holds HIST samples between read-backs)
            parts, left = [], n
            while left > 0:
                parts.append(self.run(min(left, self.HIST // N)))
                left -= min(left, self.HIST // N)
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        m, r = self.model, self.rank
        dev = m.device
        be = m._backend
        be.prepare_session(m, 1, self.caches[0])  # size the workspace as run_stack will BEFORE reading its step counter
        base = be.session_status()["steps"]       # row of the history ring the first sample of this run lands in (last stage)
        total = n * N
        nxt, prv = (r + 1) % N, (r - 1) % N
        cur = 0  # index of the activation buffer this stage computes in
        t_host = time.perf_counter()
        for t in range(total + N - 1):
            idx = t - r
            active = 0 <= idx < total
            # what the stage behind computed in this tick arrives now (it was active iff 0 <= t - prv < total)
            p_idx = t - prv
            recv_t = None
            if N > 1 and 0 <= p_idx < total:
                recv_t = self.tok[p_idx % N] if self.is_first else self.h[1 - cur]
            if active:
                j = idx % N
                self._tick(j, cur, recv_t, nxt, prv)
                self.caches[j].advance_host([1])
            else:
                self._exchange(None, nxt, recv_t, prv)  # pipeline fill / drain: nothing to compute, maybe something to receive
            if N > 1 and not self.is_first and recv_t is not None:
                cur = 1 - cur
        self.tick_host_us = (time.perf_counter() - t_host) / max(1, total + N - 1) * 1e6
        rows = (base + torch.arange(total, device=dev)) % self.HIST
        out_tok = self.hist_tok[rows, 0].reshape(n, N).contiguous()
        out_lp = self.hist_lp[rows, 0].reshape(n, N).contiguous()
        if N > 1:
            m.pp_comm.broadcast(out_tok, src=N - 1)
            m.pp_comm.broadcast(out_lp, src=N - 1)
        self.steps_done += n
        if hasattr(be, "raise_if_flagged") and dev.type == "cuda":
            be.raise_if_flagged()  # (synchronises: device-side flags of this stage become exceptions here)
        return out_tok, out_lp
