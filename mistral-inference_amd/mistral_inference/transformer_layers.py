"""Layer modules of the hot path (reference transformer_layers.py:16-169) as thin hosts over the HIP
operators.  Parameter names and shapes are the reference's (checkpoint keys load unchanged); every
`forward` dispatches to libmistral_hip -- there is no torch compute path here.

These module-level forwards exist for callers that drive a block themselves; `Transformer.forward_partial`
bypasses them and hands the whole local layer stack to the native runner (`mi_forward`)."""
from typing import Optional

import torch
from torch import nn

from . import _hip
from .args import LoraArgs, MoeArgs
from .cache import CacheView
from .moe import MoeLayer


def _no_lora(lora: Optional[LoraArgs]) -> None:
    if lora is not None:
        raise NotImplementedError(
            "un-merged LoRA layers (params.json 'lora') are outside the hot path; merge the adapter into the "
            "checkpoint first (what the reference's default CLI path does, lora.py:118-139)")


class RMSNorm(nn.Module):
    """bf16( bf16(x_f32 * rsqrt(mean(x_f32^2) + eps)) * weight )  (reference transformer_layers.py:109-120)."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _hip.rmsnorm(x, self.weight, self.eps)


class FeedForward(nn.Module):
    """w2( silu(w1 x) * w3 x ) with bf16 rounding after every step (reference transformer_layers.py:96-106):
    gate/up projections + SiLU*mul are one kernel, the down projection another."""

    def __init__(self, dim: int, hidden_dim: int, lora: Optional[LoraArgs] = None):
        super().__init__()
        _no_lora(lora)
        self.w1 = nn.Linear(dim, hidden_dim, bias=False)
        self.w2 = nn.Linear(hidden_dim, dim, bias=False)
        self.w3 = nn.Linear(dim, hidden_dim, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hid = _hip.linear(x, (self.w1.weight, self.w3.weight), _hip.EPI_SWIGLU)
        return _hip.linear(hid, (self.w2.weight,), _hip.EPI_STORE)


class Attention(nn.Module):
    """Reference transformer_layers.py:30-93.  q|k|v are ONE projection launch over the three weight
    matrices, GQA is resolved inside the attention kernels (no repeat_kv), and the three xformers masks
    are the kernels' position test."""

    def __init__(self, dim: int, n_heads: int, head_dim: int, n_kv_heads: int, lora: Optional[LoraArgs] = None):
        super().__init__()
        _no_lora(lora)
        self.n_heads, self.head_dim, self.n_kv_heads = n_heads, head_dim, n_kv_heads
        self.repeats = n_heads // n_kv_heads
        self.scale = head_dim ** -0.5
        self.wq = nn.Linear(dim, n_heads * head_dim, bias=False)
        self.wk = nn.Linear(dim, n_kv_heads * head_dim, bias=False)
        self.wv = nn.Linear(dim, n_kv_heads * head_dim, bias=False)
        self.wo = nn.Linear(n_heads * head_dim, dim, bias=False)

    def forward(self, x: torch.Tensor, freqs_cis: torch.Tensor, cache: Optional[CacheView] = None,
                mask=None) -> torch.Tensor:
        assert mask is None or cache is None
        T = x.shape[0]
        H, Hkv, Dh = self.n_heads, self.n_kv_heads, self.head_dim
        nq, nkv = H * Dh, Hkv * Dh
        cs = torch.view_as_real(freqs_cis).contiguous()  # rows already gathered by position (transformer.py:199)
        rows = torch.arange(T, dtype=torch.int32, device=x.device)
        if T <= _hip.GEMV_MAX_T:  # decode-sized: projection + RoPE in the one weight-streaming launch
            qkv = _hip.qkv_rope_kvwrite(x, self.wq.weight, self.wk.weight, self.wv.weight, Dh, cs, rows)
        else:
            qkv = _hip.linear(x, (self.wq.weight, self.wk.weight, self.wv.weight), _hip.EPI_STORE)
            _hip.rope_inplace(qkv, H, Hkv, Dh, cs, rows)
        if cache is None:
            # reference quirk: the block never forwards `mask` (transformer_layers.py:165) -> unmasked
            out = _hip.attn_prefill(qkv, H, Hkv, Dh, None, None, T, None, None, 1, T, causal=False)
        else:
            b = cache.metadata.batch
            assert b is not None
            if cache.prefill:
                out = _hip.attn_prefill(qkv, H, Hkv, Dh, cache.cache_k, cache.cache_v, cache.max_seq_len, b.q_start,
                                        b.kv_before, len(b.seqlens), b.max_q_len, causal=True)
                cache.update(qkv[:, nq:nq + nkv], qkv[:, nq + nkv:])
            else:
                cache.update(qkv[:, nq:nq + nkv], qkv[:, nq + nkv:])
                out = _hip.attn_decode(qkv, cache.cache_k, cache.cache_v, H, b.tok_pos)
        return _hip.linear(out, (self.wo.weight,), _hip.EPI_STORE)


class TransformerBlock(nn.Module):
    """Pre-norm residual block (reference transformer_layers.py:123-169)."""

    def __init__(self, dim: int, hidden_dim: int, n_heads: int, n_kv_heads: int, head_dim: int, norm_eps: float,
                 lora: Optional[LoraArgs] = None, moe: Optional[MoeArgs] = None):
        super().__init__()
        self.n_heads = n_heads
        self.dim = dim
        self.attention = Attention(dim=dim, n_heads=n_heads, head_dim=head_dim, n_kv_heads=n_kv_heads, lora=lora)
        self.attention_norm = RMSNorm(dim, eps=norm_eps)
        self.ffn_norm = RMSNorm(dim, eps=norm_eps)
        self.feed_forward: nn.Module
        if moe is not None:
            self.feed_forward = MoeLayer(
                experts=[FeedForward(dim=dim, hidden_dim=hidden_dim, lora=lora) for _ in range(moe.num_experts)],
                gate=nn.Linear(dim, moe.num_experts, bias=False), moe_args=moe)
        else:
            self.feed_forward = FeedForward(dim=dim, hidden_dim=hidden_dim, lora=lora)

    def forward(self, x: torch.Tensor, freqs_cis: torch.Tensor, cache: Optional[CacheView] = None,
                mask=None) -> torch.Tensor:
        r = self.attention.forward(self.attention_norm(x), freqs_cis, cache)
        h = x + r
        r = self.feed_forward.forward(self.ffn_norm(h))
        return h + r
