"""Pixtral vision tower (reference vision_encoder.py) as a thin host over the HIP operators.

Module and parameter names are the reference's, so `vision_encoder.*`, `vision_language_adapter.*`, `patch_merger.*` and
`pre_mm_projector_norm.*` checkpoint keys load unchanged.  All arithmetic runs in libmistral_hip (GEMMs, RMSNorm, RoPE,
attention, GELU); torch only reshapes / gathers / pads (im2col of the patch convolution, head padding, the patch-merger
permutation, the final row scatter).

How the tower maps onto the text path's kernels:
  * patch convolution (stride = kernel = patch) = one GEMM over im2col rows [n_patches, C*P*P];
  * the blocks are the text TransformerBlock with n_kv_heads = n_heads and 64-wide heads.  The attention kernels are
    built for 128-wide heads, so q/k/v are zero-padded to 128 per head (dot products and P.V are unchanged by zero
    columns), the softmax scale is passed explicitly (64^-1/2), and the 2-D RoPE table gets identity entries (cos 1,
    sin 0) for the padded pairs; rotary position = row * max_patches_per_side + column;
  * the reference builds a per-image BlockDiagonalMask but TransformerBlock.forward never hands it to the attention
    (transformer_layers.py:165), so what runs - and what is reproduced here - is ONE unmasked attention over the patches
    of all images (`causal=False` form of mi_attn_prefill);
  * adapter Linear layers with bias: the bias is folded into the GEMM as an extra K column against a column of ones,
    so bias and products are accumulated in fp32 and rounded once, like nn.Linear.
"""
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _hip
from .args import VisionEncoderArgs
from .transformer_layers import RMSNorm, TransformerBlock

PATCH_MERGE = "patch_merge"


def precompute_freqs_cis_2d(dim: int, height: int, width: int, theta: float) -> torch.Tensor:
    """complex64 [height, width, dim/2] (reference rope.py:26-51): even frequency bases turn with the row index, odd
    ones with the column index."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
    h = torch.arange(height, device=freqs.device)
    w = torch.arange(width, device=freqs.device)
    freqs_h = torch.outer(h, freqs[::2]).float()
    freqs_w = torch.outer(w, freqs[1::2]).float()
    freqs_2d = torch.cat([freqs_h[:, None, :].repeat(1, width, 1), freqs_w[None, :, :].repeat(height, 1, 1)], dim=-1)
    return torch.polar(torch.ones_like(freqs_2d), freqs_2d)


def position_meshgrid(patch_grids: List[Tuple[int, int]]) -> torch.Tensor:
    """(row, column) of every patch of every image, images concatenated (reference vision_encoder.py:12-29)."""
    return torch.cat([torch.stack(torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij"), dim=-1).reshape(-1, 2)
                      for gh, gw in patch_grids])


def _linear_bias(x: torch.Tensor, lin: nn.Linear, cache: dict, key: str) -> torch.Tensor:
    """bf16(x @ W^T + b) with the bias inside the fp32 accumulation: K is extended by 8 columns (GEMM K granularity), the
    first holding b against a column of ones."""
    if lin.bias is None:
        return _hip.linear(x, (lin.weight,), _hip.EPI_STORE)
    if key not in cache:
        pad = torch.zeros((lin.weight.shape[0], 8), dtype=lin.weight.dtype, device=lin.weight.device)
        pad[:, 0] = lin.bias
        cache[key] = torch.cat([lin.weight, pad], dim=1).contiguous()
    ones = torch.zeros((x.shape[0], 8), dtype=x.dtype, device=x.device)
    ones[:, 0] = 1
    return _hip.linear(torch.cat([x, ones], dim=1).contiguous(), (cache[key],), _hip.EPI_STORE)


class VisionTransformerBlocks(nn.Module):
    def __init__(self, args: VisionEncoderArgs):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        for _ in range(args.num_hidden_layers):
            self.layers.append(TransformerBlock(dim=args.hidden_size, hidden_dim=args.intermediate_size,
                                                n_heads=args.num_attention_heads, n_kv_heads=args.num_attention_heads,
                                                head_dim=args.hidden_size // args.num_attention_heads, norm_eps=1e-5))


class VisionTransformer(nn.Module):
    def __init__(self, args: VisionEncoderArgs):
        super().__init__()
        self.args = args
        self.patch_conv = nn.Conv2d(in_channels=args.num_channels, out_channels=args.hidden_size,
                                    kernel_size=args.patch_size, stride=args.patch_size, bias=False)
        self.ln_pre = RMSNorm(args.hidden_size, eps=1e-5)
        self.transformer = VisionTransformerBlocks(args)
        head_dim = args.hidden_size // args.num_attention_heads
        assert head_dim % 2 == 0, "ROPE requires even head_dim"
        self._head_dim = head_dim  # bf16: the MFMA attention takes 128 (64 zero-padded); fp16 / fp32: any head dim <= 256
        self._rope_cs: Optional[torch.Tensor] = None
        self._conv_w: Optional[torch.Tensor] = None  # K-padded image of patch_conv.weight (rebuilt after a reload)

    @property
    def max_patches_per_side(self) -> int:
        return self.args.image_size // self.args.patch_size

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def freqs_cis(self) -> torch.Tensor:
        a = self.args
        side = self.max_patches_per_side
        return precompute_freqs_cis_2d(a.hidden_size // a.num_attention_heads, side, side, a.rope_theta).to(self.device)

    def _rope_table(self, padded: bool = True) -> torch.Tensor:
        """fp32 [side*side, 64, 2]: (cos, sin) of the head's real pairs, identity for the zero-padded ones (bf16 kernels);
        `padded=False`: [side*side, head_dim/2, 2], the real pairs only (generic kernels: heads keep their own width)."""
        key = "p" if padded else "r"
        if self._rope_cs is None or self._rope_cs[0] != (key, self.device):
            side = self.max_patches_per_side
            real = torch.view_as_real(self.freqs_cis).reshape(side * side, -1, 2)  # [side^2, head_dim/2, 2]
            if padded:
                cs = torch.zeros((side * side, 64, 2), dtype=torch.float32, device=self.device)
                cs[:, :, 0] = 1.0
                cs[:, : real.shape[1]] = real
            else:
                cs = real.to(torch.float32)
            self._rope_cs = ((key, self.device), cs.contiguous())
        return self._rope_cs[1]

    def forward(self, images: List[torch.Tensor]) -> torch.Tensor:
        """images: list of [C, H, W] tensors (H, W multiples of the patch size) -> [sum of patches, hidden]."""
        a = self.args
        P, H = a.patch_size, a.num_attention_heads
        dh = a.hidden_size // H
        dev = self.device
        grids = [(img.shape[1] // P, img.shape[2] // P) for img in images]
        side = self.max_patches_per_side
        assert all(gh <= side and gw <= side for gh, gw in grids), (
            f"image larger than image_size={a.image_size}: the 2-D rotary table has {side} positions per axis")
        rows = []
        for img in images:  # im2col in nn.Conv2d's weight order (c, py, px), patches row-major over the grid
            C_, Hh, Ww = img.shape
            x = img.to(dev).unfold(1, P, P).unfold(2, P, P)
            rows.append(x.permute(1, 2, 0, 3, 4).reshape((Hh // P) * (Ww // P), C_ * P * P))
        patches = torch.cat(rows)
        w_flat = self.patch_conv.weight.view(a.hidden_size, -1)
        kpad = (-patches.shape[1]) % 8  # the GEMM's K granularity (C*P*P = 588 for 14-pixel patches): zero columns
        if kpad:
            if self._conv_w is None or self._conv_w.device != dev:
                self._conv_w = torch.nn.functional.pad(w_flat, (0, kpad)).contiguous()
            patches, w_flat = torch.nn.functional.pad(patches, (0, kpad)), self._conv_w
        x = _hip.linear(patches.contiguous(), (w_flat,), _hip.EPI_STORE)
        x = _hip.rmsnorm(x, self.ln_pre.weight, 1e-5)
        T = x.shape[0]
        pos = position_meshgrid(grids)
        pos_id = (pos[:, 0] * self.max_patches_per_side + pos[:, 1]).to(device=dev, dtype=torch.int32)
        generic = x.dtype != torch.bfloat16   # fp16 / fp32 storage: csrc/generic.hip takes the heads as they are
        if not generic and dh not in (64, 128):
            raise NotImplementedError(f"vision head_dim {dh}: the bf16 attention kernels take 128 (or 64, zero-padded)")
        cs = self._rope_table(padded=not generic)
        for blk in self.transformer.layers:
            at = blk.attention
            xn = _hip.rmsnorm(x, blk.attention_norm.weight, 1e-5)
            qkv = _hip.linear(xn, (at.wq.weight, at.wk.weight, at.wv.weight), _hip.EPI_STORE)  # [T, 3*H*dh]
            if generic:
                _hip.rope_inplace(qkv, H, H, dh, cs, pos_id)
                att = _hip.attn_prefill(qkv, H, H, dh, None, None, T, None, None, 1, T, causal=False, softmax_scale=dh ** -0.5)
            else:
                if dh != 128:  # zero-pad every head to the kernels' 128 columns
                    pad = torch.zeros((T, 3 * H, 128), dtype=qkv.dtype, device=dev)
                    pad[:, :, :dh] = qkv.view(T, 3 * H, dh)
                    qkv = pad.view(T, 3 * H * 128)
                _hip.rope_inplace(qkv, H, H, 128, cs, pos_id)
                att = _hip.attn_prefill(qkv, H, H, 128, None, None, T, None, None, 1, T, causal=False,
                                        softmax_scale=dh ** -0.5)
                if dh != 128:
                    att = att.view(T, H, 128)[:, :, :dh].reshape(T, H * dh).contiguous()
            x = _hip.linear(att, (at.wo.weight,), _hip.EPI_RESIDUAL, residual=x)
            ff = blk.feed_forward
            xn = _hip.rmsnorm(x, blk.ffn_norm.weight, 1e-5)
            hid = _hip.linear(xn, (ff.w1.weight, ff.w3.weight), _hip.EPI_SWIGLU)
            x = _hip.linear(hid, (ff.w2.weight,), _hip.EPI_RESIDUAL, residual=x)
        return x


class VisionLanguageAdapter(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, bias: bool = True):
        super().__init__()
        self.w_in = nn.Linear(in_dim, out_dim, bias=bias)
        self.gelu = nn.GELU()
        self.w_out = nn.Linear(out_dim, out_dim, bias=bias)
        self._aug: dict = {}

    def invalidate(self) -> None:
        self._aug = {}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = _linear_bias(x, self.w_in, self._aug, "in")
        _hip.gelu_(h)
        return _linear_bias(h, self.w_out, self._aug, "out")


class PatchMerger(nn.Module):
    """Learned merging of spatial_merge_size ** 2 patches (reference vision_encoder.py:147-204)."""

    def __init__(self, vision_encoder_dim: int, spatial_merge_size: int) -> None:
        super().__init__()
        self.spatial_merge_size = spatial_merge_size
        self.mlp_input_dim = vision_encoder_dim * (spatial_merge_size ** 2)
        self.merging_layer = nn.Linear(self.mlp_input_dim, vision_encoder_dim, bias=False)

    def permute(self, x: torch.Tensor, image_sizes: List[Tuple[int, int]]) -> torch.Tensor:
        """Every s x s block of patches becomes one row, features ordered (d, dy, dx) as F.unfold orders them."""
        s, d = self.spatial_merge_size, x.shape[-1]
        rows, o = [], 0
        for gh, gw in image_sizes:
            g = x[o:o + gh * gw].view(gh // s, s, gw // s, s, d).permute(0, 2, 4, 1, 3)
            rows.append(g.reshape((gh // s) * (gw // s), d * s * s))
            o += gh * gw
        return torch.cat(rows).contiguous()

    def forward(self, x: torch.Tensor, image_sizes: List[Tuple[int, int]]) -> torch.Tensor:
        assert sum(h * w for h, w in image_sizes) == len(x), f"{sum(h * w for h, w in image_sizes)} != {len(x)}"
        return _hip.linear(self.permute(x, image_sizes), (self.merging_layer.weight,), _hip.EPI_STORE)
