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
    bias = None
    if attn_bias is not None:
        bias = attn_bias.materialize((b, h, mq, mk), dtype=torch.float32, device=query.device)
        # xformers never READS a key no query of the call can see (the padded tail of every cache row under
        # BlockDiagonalCausalWithOffsetPaddedKeysMask, reference cache.py:249-254).  Those rows of the reference's
        # `torch.empty` cache (cache.py:163-167) hold arbitrary bits - NaN/Inf included - and `NaN + -inf` or `0 * NaN`
        # would leak them into the result, so dead key columns are zeroed before they enter any arithmetic.
        dead = torch.isneginf(bias).all(dim=-2)  # [b, h, mk]
        kk = kk.masked_fill(dead.unsqueeze(-1), 0.0)
        v = v.masked_fill(dead.unsqueeze(-1), 0.0)
    s = torch.matmul(q, kk.transpose(-1, -2)) * sc
    if bias is not None:
        s = (s + bias).masked_fill(torch.isneginf(bias), float("-inf"))
    p_ = torch.softmax(s, dim=-1)
    o = torch.matmul(p_, v)
    return o.permute(0, 2, 1, 3).contiguous().to(query.dtype)
