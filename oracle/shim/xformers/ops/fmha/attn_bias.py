"""Restatement of the xformers.ops.fmha.attn_bias classes used at reference cache.py:238-254.

Every mask materialises to an additive fp32 bias [Mq, Mk] (0 = visible, -inf = hidden).
Window conventions follow xformers 0.0.26: `_window_size = W` keeps key j for query i iff
i - W < j <= i (causal local) or, bottom-right aligned, i + shift - W < j <= i + shift with
shift = nk - nq of the block.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch


class AttentionBias:
    def materialize(self, shape, dtype=torch.float32, device="cpu") -> torch.Tensor:  # pragma: no cover
        raise NotImplementedError


def _starts(lens: Sequence[int]) -> List[int]:
    out, acc = [], 0
    for n in lens:
        out.append(acc)
        acc += n
    return out


@dataclass
class BlockDiagonalMask(AttentionBias):
    q_seqlen: List[int]
    kv_seqlen: List[int]
    causal: bool = False          # top-left aligned causal inside each block
    window: Optional[int] = None  # local attention window
    bottomright: bool = False     # window/causal diagonal aligned to the bottom-right corner

    @classmethod
    def from_seqlens(cls, q_seqlen: Sequence[int], kv_seqlen: Optional[Sequence[int]] = None):
        q = list(q_seqlen)
        kv = list(kv_seqlen) if kv_seqlen is not None else list(q)
        assert len(q) == len(kv)
        return cls(q_seqlen=q, kv_seqlen=kv)

    def make_local_attention_from_bottomright(self, window_size: int) -> "BlockDiagonalMask":
        return BlockDiagonalMask(self.q_seqlen, self.kv_seqlen, causal=True, window=window_size, bottomright=True)

    def materialize(self, shape, dtype=torch.float32, device="cpu") -> torch.Tensor:
        mq, mk = sum(self.q_seqlen), sum(self.kv_seqlen)
        assert tuple(shape[-2:]) == (mq, mk), (shape, mq, mk)
        bias = torch.full((mq, mk), float("-inf"), dtype=dtype, device=device)
        for qs, nq, ks, nk in zip(_starts(self.q_seqlen), self.q_seqlen, _starts(self.kv_seqlen), self.kv_seqlen):
            i = torch.arange(nq, device=device)[:, None]
            j = torch.arange(nk, device=device)[None, :]
            shift = (nk - nq) if self.bottomright else 0
            vis = torch.ones(nq, nk, dtype=torch.bool, device=device)
            if self.causal:
                vis &= j <= i + shift
            if self.window is not None:
                vis &= j > i + shift - self.window
            blk = torch.zeros(nq, nk, dtype=dtype, device=device)
            blk[~vis] = float("-inf")
            bias[qs : qs + nq, ks : ks + nk] = blk
        return bias


@dataclass
class BlockDiagonalCausalMask(BlockDiagonalMask):
    causal: bool = True

    @classmethod
    def from_seqlens(cls, q_seqlen: Sequence[int], kv_seqlen: Optional[Sequence[int]] = None):
        q = list(q_seqlen)
        kv = list(kv_seqlen) if kv_seqlen is not None else list(q)
        return cls(q_seqlen=q, kv_seqlen=kv, causal=True)

    def make_local_attention(self, window_size: int) -> "BlockDiagonalMask":
        return BlockDiagonalMask(self.q_seqlen, self.kv_seqlen, causal=True, window=window_size, bottomright=False)


@dataclass
class BlockDiagonalCausalWithOffsetPaddedKeysMask(AttentionBias):
    """Decode mask: sequence b owns key slots [b*kv_padding, b*kv_padding + kv_seqlen[b]).

    Causality is bottom-right aligned; with one query per sequence every valid key is visible.
    """

    q_seqlen: List[int]
    kv_padding: int
    kv_seqlen: List[int]

    @classmethod
    def from_seqlens(cls, q_seqlen: Sequence[int], kv_padding: int, kv_seqlen: Sequence[int]):
        q, kv = list(q_seqlen), list(kv_seqlen)
        assert len(q) == len(kv)
        assert all(0 <= n <= kv_padding for n in kv)
        return cls(q_seqlen=q, kv_padding=kv_padding, kv_seqlen=kv)

    def materialize(self, shape, dtype=torch.float32, device="cpu") -> torch.Tensor:
        mq, mk = sum(self.q_seqlen), len(self.kv_seqlen) * self.kv_padding
        assert tuple(shape[-2:]) == (mq, mk), (shape, mq, mk)
        bias = torch.full((mq, mk), float("-inf"), dtype=dtype, device=device)
        for b, (qs, nq, nk) in enumerate(zip(_starts(self.q_seqlen), self.q_seqlen, self.kv_seqlen)):
            i = torch.arange(nq, device=device)[:, None]
            j = torch.arange(nk, device=device)[None, :]
            vis = j <= i + (nk - nq)
            blk = torch.zeros(nq, nk, dtype=dtype, device=device)
            blk[~vis] = float("-inf")
            bias[qs : qs + nq, b * self.kv_padding : b * self.kv_padding + nk] = blk
        return bias
