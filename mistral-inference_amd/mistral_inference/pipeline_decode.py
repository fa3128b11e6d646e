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
    n greedy tokens for every sequence and returns (tokens int64 [n, N], logprobs fp32 [n, N]) on every stage."""

    def __init__(self, model: Transformer, caches: List[BufferCache], first_tokens: torch.Tensor):
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
        self.bufs = [GreedyBuffers(tok=self.tok[j], lp=torch.zeros(1, dtype=torch.float32, device=dev),
                                   hist_tok=torch.zeros((1, 1), dtype=torch.long, device=dev),
                                   hist_lp=torch.zeros((1, 1), dtype=torch.float32, device=dev)) for j in range(self.n_seq)]
        self.logits = torch.empty((1, model.vocab_size), dtype=torch.float32, device=dev) if self.is_last else None
        self.h = [torch.empty((1, model.args.dim), dtype=model.dtype, device=dev) for _ in range(2)]  # in flight / being filled
        self.steps_done = 0
        self.tick_host_us = 0.0  # host time per tick of the last run() (enqueue cost of one stage call + one grouped exchange)

    def run(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        m, N, r = self.model, self.world, self.rank
        dev = m.device
        be = m._backend
        out_tok = torch.zeros((n, N), dtype=torch.long, device=dev)
        out_lp = torch.zeros((n, N), dtype=torch.float32, device=dev)
        total = n * N
        nxt, prv = (r + 1) % N, (r - 1) % N
        cur = 0  # index of the activation buffer this stage computes in
        t_host = time.perf_counter()
        for t in range(total + N - 1):
            idx = t - r
            active = 0 <= idx < total
            send_t = None
            if active:
                j, k = idx % N, idx // N
                cache = self.caches[j]
                meta = cache.batch_metadata([1])
                assert meta.branch == _hip.BRANCH_DECODE
                h = self.h[cur]
                be.run_stack(m, h, self.tok[j] if self.is_first else None, meta, cache, self.logits,
                             greedy=self.bufs[j] if self.is_last else None)
                cache.advance_host([1])
                if self.is_last:
                    out_tok[k, j].copy_(self.tok[j][0])
                    out_lp[k, j].copy_(self.bufs[j].lp[0])
                    send_t = self.tok[j]      # the sample: 8 bytes back to stage 0
                else:
                    send_t = h                # [1, dim] to the next stage
            # what the stage behind computed in this tick arrives now (it was active iff 0 <= t - prv < total)
            p_idx = t - prv
            recv_t = None
            if N > 1 and 0 <= p_idx < total:
                recv_t = self.tok[p_idx % N] if self.is_first else self.h[1 - cur]
            if N > 1:
                m.pp_comm.exchange(send_t, nxt, recv_t, prv)
                if not self.is_first and recv_t is not None:
                    cur = 1 - cur
        self.tick_host_us = (time.perf_counter() - t_host) / max(1, total + N - 1) * 1e6
        if N > 1:
            m.pp_comm.broadcast(out_tok, src=N - 1)
            m.pp_comm.broadcast(out_lp, src=N - 1)
        self.steps_done += n
        if hasattr(be, "raise_if_flagged") and dev.type == "cuda":
            be.raise_if_flagged()  # (synchronises: device-side flags of this stage become exceptions here)
        return out_tok, out_lp
