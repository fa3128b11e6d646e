"""Rotating K/V buffer cache (reference cache.py:13-263), device-resident and sync-free.

Same constructor, attributes and methods as the reference's `BufferCache` (exported also under the
historical name `RotatingBufferCache`), same ring semantics: layer l keeps the last W_l tokens of each
sequence at slot `pos % W_l` of row b of `cache_k[l]` / `cache_v[l]` ([max_batch, W_l, n_kv_heads,
head_dim] - the reference's shape; the STORAGE behind it is head-major, see `BufferCache.__init__`).  What differs is how a
forward learns about it:

* the reference rebuilds five small tensors + an xformers mask per LAYER per forward on the host and
  syncs the device several times (cache.py:217,246,253); here one int32 metadata block per FORWARD is
  uploaded for prefill, and nothing at all for decode (positions are derived on the device from
  `kv_seqlens`, which a kernel also advances);
* a host mirror of `kv_seqlens` removes every `.tolist()` / `.item()`;
* no mask objects exist: visibility is `qpos - W < kpos <= qpos`, evaluated inside the attention
  kernels.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import torch

from . import _hip

SlidingWindow = Union[None, int, List[Optional[int]]]


def get_cache_sizes(n_layers: int, max_seq_len: int, sliding_window: SlidingWindow) -> List[int]:
    """Per-layer ring length (reference cache.py:13-24)."""
    if sliding_window is None:
        return [max_seq_len] * n_layers
    if isinstance(sliding_window, int):
        return [sliding_window] * n_layers
    assert isinstance(sliding_window, list), f"Expected list, got {type(sliding_window)}"
    assert n_layers % len(sliding_window) == 0, (
        f"Expected n_layers % len(sliding_window) == 0, got {n_layers} % {len(sliding_window)}")
    pattern = [max_seq_len if w is None else w for w in sliding_window]
    return pattern * (n_layers // len(sliding_window))


@dataclass
class BatchMetadata:
    """What the kernels need for one forward: int32 device arrays (views into one block)."""

    branch: int
    seqlens: List[int]
    max_q_len: int
    q_start: torch.Tensor    # [B+1]
    kv_before: torch.Tensor  # [B]
    tok_seq: torch.Tensor    # [T]
    tok_pos: torch.Tensor    # [T]


@dataclass
class CacheInputMetadata:
    """Field-compatible with the reference dataclass (cache.py:27-51); `mask` is always None because
    masking lives inside the attention kernels.  `batch` carries the device-side form."""

    positions: torch.Tensor
    to_cache_mask: torch.Tensor
    cached_elements: torch.Tensor
    cache_positions: torch.Tensor
    prefill: bool
    mask: None
    seqlens: List[int]
    batch: Optional[BatchMetadata] = None


def interleave_list(l1: List[torch.Tensor], l2: List[torch.Tensor]) -> List[torch.Tensor]:
    assert len(l1) == len(l2)
    return [v for pair in zip(l1, l2) for v in pair]


def unrotate(cache: torch.Tensor, seqlen: int) -> torch.Tensor:
    """Ring [W, H, D] -> its valid tokens in position order (reference cache.py:59-67)."""
    assert cache.ndim == 3
    W = cache.shape[0]
    if seqlen < W:
        return cache[:seqlen]
    cut = seqlen % W
    return cache if cut == 0 else torch.cat([cache[cut:], cache[:cut]], dim=0)


class CacheView:
    """One layer's ring plus this forward's metadata (reference cache.py:70-137)."""

    def __init__(self, cache_k: torch.Tensor, cache_v: torch.Tensor, metadata: CacheInputMetadata,
                 kv_seqlens: torch.Tensor):
        self.cache_k = cache_k
        self.cache_v = cache_v
        self.kv_seqlens = kv_seqlens
        self.metadata = metadata

    def update(self, xk: torch.Tensor, xv: torch.Tensor) -> None:
        """Ring write of the last W tokens of each sequence (reference cache.py:83-92), one kernel, no
        boolean-mask indexing (which costs the reference two host syncs per layer)."""
        b = self.metadata.batch
        assert b is not None
        T = xk.shape[0]
        _hip.kv_write(self.cache_k, self.cache_v, xk.reshape(T, -1), xv.reshape(T, -1), b.tok_seq, b.tok_pos, b.q_start)

    def interleave_kv(self, xk: torch.Tensor, xv: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per sequence: [its cached tokens in position order ++ its new tokens], sequences concatenated (reference
        cache.py:94-117).  The kernels never need this tensor (the prefill attention reads the ring and the activations
        in place); it exists for callers that drive the layers themselves.  Pure data movement: one gather per tensor
        from [ring rows ++ new rows] with an index list assembled on the host."""
        assert xk.ndim == xv.ndim == 3 and xk.shape == xv.shape  # (sum of new tokens, H, D)
        new = self.metadata.seqlens
        seen = [int(s) for s in self.kv_seqlens.tolist()]
        assert len(new) == len(seen), f"Batch size is {len(seen)}, got {len(new)}"
        if all(s == 0 for s in seen):
            return xk, xv  # nothing cached yet
        B, W = self.cache_k.shape[0], self.cache_k.shape[1]
        index: List[int] = []
        row = 0
        for b, (p, n) in enumerate(zip(seen, new)):
            index += [b * W + q % W for q in range(p - min(p, W), p)]   # ring slots of the surviving cached positions
            index += list(range(B * W + row, B * W + row + n))         # then this sequence's new rows
            row += n
        idx = torch.tensor(index, dtype=torch.long, device=xk.device)
        flat = lambda c, x: torch.cat([c.reshape(B * W, *c.shape[2:]), x], dim=0).index_select(0, idx)  # noqa: E731
        return flat(self.cache_k, xk), flat(self.cache_v, xv)

    @property
    def max_seq_len(self) -> int:
        return self.cache_k.shape[1]

    @property
    def key(self) -> torch.Tensor:
        return self.cache_k[: len(self.kv_seqlens)]

    @property
    def value(self) -> torch.Tensor:
        return self.cache_v[: len(self.kv_seqlens)]

    @property
    def prefill(self) -> bool:
        return self.metadata.prefill

    @property
    def mask(self) -> None:
        return None


class BufferCache:
    """Rectangular rotating K/V cache, caller-owned (created by `generate()`, reference generate.py:69-78)."""

    def __init__(self, n_layers: int, max_batch_size: int, max_seq_len: int, n_kv_heads: int, head_dim: int,
                 sliding_window: SlidingWindow = None, *, device: Union[None, str, torch.device] = None,
                 dtype: Optional[torch.dtype] = None):
        self.max_seq_len = max_seq_len
        self.n_kv_heads = n_kv_heads
        self.head_dim = head_dim
        self.n_layers = n_layers
        self.max_batch_size = max_batch_size
        self.cache_sizes: List[int] = get_cache_sizes(n_layers, max_seq_len, sliding_window)
        assert len(self.cache_sizes) == n_layers, f"Expected {n_layers} cache sizes, got {len(self.cache_sizes)}"
        # The reference allocates on the host and moves later (cache.py:163-167,186-191); passing `device`
        # allocates in HBM directly (what this package's generate() does).
        kw = dict(device=device, dtype=dtype)
        self.cache_k: Dict[int, torch.Tensor] = {}
        self.cache_v: Dict[int, torch.Tensor] = {}
        # Layout in HBM: head-major [max_batch, n_kv_heads, W, head_dim] - the slots of one kv head are
        # contiguous, so what a decode work item (kv head, range of slots) reads is one run - exposed in the reference's SHAPE
        # [max_batch, W, n_kv_heads, head_dim] (cache.py:163-167) as a permuted view: indexing, `torch.equal`, `CacheView.key`
        # behave as in the reference; only `.view()` / `.is_contiguous()` can tell.  MI_KV_LAYOUT=0 allocates the reference's
        # layout (A/B runs, tests); the kernels take both (`_hip.kv_layout_of`).
        self.head_major = os.environ.get("MI_KV_LAYOUT", "1") != "0"
        for i, w in enumerate(self.cache_sizes):
            self.cache_k[i] = self._alloc(max_batch_size, w, **kw)
            self.cache_v[i] = self._alloc(max_batch_size, w, **kw)
        self.kv_seqlens: Optional[torch.Tensor] = None  # device int64 [B], as in the reference
        self._seen: Optional[List[int]] = None          # host mirror of kv_seqlens
        self._decode_meta: Optional[torch.Tensor] = None
        self._ptr_tables = None

    def _alloc(self, B: int, w: int, **kw) -> torch.Tensor:
        if self.head_major:
            return torch.empty((B, self.n_kv_heads, w, self.head_dim), **kw).permute(0, 2, 1, 3)
        return torch.empty((B, w, self.n_kv_heads, self.head_dim), **kw)

    @property
    def kv_layout(self) -> int:
        """`_hip.KV_SLOT_MAJOR` / `_hip.KV_HEAD_MAJOR` of every ring (checked: a caller may have replaced tensors)."""
        if self.n_layers == 0:
            return _hip.KV_SLOT_MAJOR
        lay = {_hip.kv_layout_of(t) for d in (self.cache_k, self.cache_v) for i, t in d.items() if self.cache_sizes[i] > 1 and self.n_kv_heads > 1}
        assert len(lay) <= 1, "K/V rings of one cache in different layouts"
        return lay.pop() if lay else _hip.KV_SLOT_MAJOR

    # ---- reference API ----------------------------------------------------------------------
    def get_view(self, layer_id: int, metadata: CacheInputMetadata) -> CacheView:
        assert self.kv_seqlens is not None
        return CacheView(self.cache_k[layer_id], self.cache_v[layer_id], metadata, self.kv_seqlens)

    def reset(self) -> None:
        self.kv_seqlens = None
        self._seen = None

    def init_kvseqlens(self, batch_size: int) -> None:
        self.kv_seqlens = torch.zeros((batch_size,), device=self.device, dtype=torch.long)
        self._seen = [0] * batch_size

    @property
    def device(self) -> torch.device:
        return self.cache_k[0].device

    def to(self, device: Union[str, torch.device], dtype: torch.dtype) -> "BufferCache":
        def move(t: torch.Tensor) -> torch.Tensor:  # (keeps the layout: a head-major ring stays a permuted view of dense storage)
            if _hip.kv_layout_of(t) == _hip.KV_HEAD_MAJOR:
                return t.permute(0, 2, 1, 3).to(device=device, dtype=dtype).contiguous().permute(0, 2, 1, 3)
            return t.to(device=device, dtype=dtype)
        for i in range(self.n_layers):
            self.cache_k[i] = move(self.cache_k[i])
            self.cache_v[i] = move(self.cache_v[i])
        self._ptr_tables = None
        self._decode_meta = None
        if self.kv_seqlens is not None:
            self.kv_seqlens = self.kv_seqlens.to(device)
        return self

    def update_seqlens(self, seqlens: List[int]) -> None:
        assert self.kv_seqlens is not None and self._seen is not None
        self.kv_seqlens += torch.tensor(seqlens, device=self.device, dtype=torch.long)
        self._seen = [p + s for p, s in zip(self._seen, seqlens)]

    def get_input_metadata(self, seqlens: List[int]) -> List[CacheInputMetadata]:
        """Per-layer metadata with the reference's field meanings (cache.py:197-263).  Provided for API
        compatibility and tests; `Transformer` itself only needs `batch_metadata()`."""
        batch = self.batch_metadata(seqlens)
        seen = self._seen
        assert seen is not None
        dev = self.device
        if batch.branch == _hip.BRANCH_DECODE:
            # `batch_metadata` hands out views of the block the decode-prep kernel of `mi_forward` fills on the device.
            # A caller that drives the layers itself never runs that kernel: give it the same arrays from the host mirror.
            B = len(seqlens)
            blob = torch.tensor(list(range(B + 1)) + list(seen) + list(range(B)) + list(seen), dtype=torch.int32).to(dev)
            batch = BatchMetadata(_hip.BRANCH_DECODE, seqlens, 1, blob[: B + 1], blob[B + 1: 2 * B + 1],
                                  blob[2 * B + 1: 3 * B + 1], blob[3 * B + 1:])
        out: List[CacheInputMetadata] = []
        tok_b = [b for b, s in enumerate(seqlens) for _ in range(s)]
        tok_i = [i for s in seqlens for i in range(s)]
        pos = [seen[b] + i for b, i in zip(tok_b, tok_i)]
        for w in self.cache_sizes:
            keep = [i >= seqlens[b] - w for b, i in zip(tok_b, tok_i)]
            out.append(CacheInputMetadata(
                positions=torch.tensor(pos, device=dev, dtype=torch.long),
                to_cache_mask=torch.tensor(keep, device=dev, dtype=torch.bool),
                cached_elements=torch.tensor([min(s, w) for s in seqlens], device=dev, dtype=torch.long),
                cache_positions=torch.tensor([p % w + b * w for p, b, k in zip(pos, tok_b, keep) if k], device=dev,
                                             dtype=torch.long),
                prefill=batch.branch == _hip.BRANCH_PREFILL, mask=None, seqlens=seqlens, batch=batch))
        return out

    # ---- device-side form --------------------------------------------------------------------
    def batch_metadata(self, seqlens: List[int]) -> BatchMetadata:
        """Branch selection of reference cache.py:236-254 + the int32 arrays the kernels read."""
        assert len(seqlens) > 0, seqlens
        if self.kv_seqlens is None:
            self.init_kvseqlens(len(seqlens))
        seen = self._seen
        assert seen is not None
        assert len(seqlens) == len(seen), (
            f"Batch size is {len(seen)}, got {len(seqlens)}, did you forget to reset cache?")
        B, T = len(seqlens), sum(seqlens)
        first_prefill = seen[0] == 0
        if first_prefill:
            assert all(p == 0 for p in seen), seen
        if not first_prefill and not any(s > 1 for s in seqlens):
            # decode: metadata is produced on the device from kv_seqlens by the first kernel of the step
            if self._decode_meta is None or self._decode_meta.numel() < 4 * B + 1:
                self._decode_meta = torch.zeros(4 * self.max_batch_size + 1, dtype=torch.int32, device=self.device)
            m = self._decode_meta
            return BatchMetadata(_hip.BRANCH_DECODE, seqlens, 1, m[: B + 1], m[B + 1: 2 * B + 1],
                                 m[2 * B + 1: 3 * B + 1], m[3 * B + 1: 4 * B + 1])
        q_start = [0]
        for s in seqlens:
            q_start.append(q_start[-1] + s)
        tok_seq = [b for b, s in enumerate(seqlens) for _ in range(s)]
        tok_pos = [seen[b] + i for b, s in enumerate(seqlens) for i in range(s)]
        blob = torch.tensor(q_start + list(seen) + tok_seq + tok_pos, dtype=torch.int32).to(self.device, non_blocking=True)
        o1, o2, o3 = B + 1, 2 * B + 1, 2 * B + 1 + T
        return BatchMetadata(_hip.BRANCH_PREFILL, seqlens, max(seqlens), blob[:o1], blob[o1:o2], blob[o2:o3], blob[o3:])

    def advance_host(self, seqlens: List[int]) -> None:
        """Host mirror only (decode: the device copy was advanced by the decode-prep kernel)."""
        assert self._seen is not None
        self._seen = [p + s for p, s in zip(self._seen, seqlens)]

    def pointer_tables(self):
        """ctypes arrays (host) of the per-layer ring pointers and sizes for mi_batch_t, and the rings' layout code."""
        if self._ptr_tables is None:
            import ctypes as C
            dt = self.cache_k[0].dtype  # (the model checks it against its own storage dtype: HipStackBackend.run_stack)
            ks = _hip.ptr_array([_hip.dev_ptr(self.cache_k[i], dt) for i in range(self.n_layers)])
            vs = _hip.ptr_array([_hip.dev_ptr(self.cache_v[i], dt) for i in range(self.n_layers)])
            ws = (C.c_int32 * self.n_layers)(*self.cache_sizes)
            self._ptr_tables = (ks, vs, ws, self.kv_layout)  # (the layout is checked once per table build, not per forward)
        return self._ptr_tables


RotatingBufferCache = BufferCache  # name used by earlier releases and by BASELINE.json's north_star
