"""RoPE tables (reference rope.py:6-23).  The table is built on the host exactly as the reference builds
it (complex64 polar form); the kernels consume its real view [pos, Dh/2, (cos, sin)] -- no in-kernel
sin/cos."""
from typing import Tuple

import torch

from . import _hip


def precompute_freqs_cis(dim: int, end: int, theta: float) -> torch.Tensor:
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    t = torch.arange(end, device=freqs.device)
    angles = torch.outer(t, freqs).float()
    return torch.polar(torch.ones_like(angles), angles)  # complex64


def apply_rotary_emb(xq: torch.Tensor, xk: torch.Tensor, freqs_cis: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """xq [T, H, Dh], xk [T, Hkv, Dh] (bf16, device), freqs_cis [T, Dh/2] complex64 = table rows of the
    tokens' positions.  Returns rotated copies (the fused decode kernels rotate inside the GEMV epilogue)."""
    T, H, Dh = xq.shape
    Hkv = xk.shape[1]
    buf = torch.cat([xq.reshape(T, H * Dh), xk.reshape(T, Hkv * Dh)], dim=1).contiguous()
    cs = torch.view_as_real(freqs_cis).contiguous()
    pos = torch.arange(T, dtype=torch.int32, device=xq.device)
    _hip.rope_inplace(buf, H, Hkv, Dh, cs, pos)
    return buf[:, : H * Dh].reshape(T, H, Dh), buf[:, H * Dh:].reshape(T, Hkv, Dh)
