"""`memory_efficient_attention` stand-in (reference call site transformer_layers.py:87-88)."""
from typing import Optional

import torch

from . import attn_bias
from .attn_bias import AttentionBias


def memory_efficient_attention(query, key, value, attn_bias: Optional[AttentionBias] = None, p: float = 0.0,
                               scale: Optional[float] = None):
    """query [1, Mq, H, K], key/value [1, Mk, H, K] -> [1, Mq, H, K] in query.dtype.

    softmax(q k^T * K^-0.5 + bias) v with fp32 scores / softmax / accumulation.
    """
    assert query.ndim == key.ndim == value.ndim == 4 and p == 0.0
    b, mq, h, k = query.shape
    mk = key.shape[1]
    sc = (k ** -0.5) if scale is None else scale
    q = query.float().permute(0, 2, 1, 3)
    kk = key.float().permute(0, 2, 1, 3)
    v = value.float().permute(0, 2, 1, 3)
    s = torch.matmul(q, kk.transpose(-1, -2)) * sc
    if attn_bias is not None:
        bias = attn_bias.materialize((b, h, mq, mk), dtype=torch.float32, device=query.device)
        s = s + bias
    p_ = torch.softmax(s, dim=-1)
    o = torch.matmul(p_, v)
    return o.permute(0, 2, 1, 3).contiguous().to(query.dtype)
