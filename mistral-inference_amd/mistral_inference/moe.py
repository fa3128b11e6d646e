"""Sparse MoE layer (reference moe.py:10-32) on the gfx950 kernels.

Module-level forward (used when a block is driven layer by layer); `Transformer.forward_partial` runs the
same kernels from the native layer-stack runner.  At decode sizes (<= 8 tokens) the router, the selected
experts' gate/up GEMVs and the down-projection + weighted bf16 combine are three launches with no host
sync; the reference needs `num_experts` `torch.where` syncs per layer (moe.py:30)."""
from typing import List

import torch
from torch import nn

from . import _hip
from .args import MoeArgs  # noqa: F401  (re-exported like the reference module)


class MoeLayer(nn.Module):
    def __init__(self, experts: List[nn.Module], gate: nn.Module, moe_args: MoeArgs):
        super().__init__()
        assert len(experts) > 0
        self.experts = nn.ModuleList(experts)
        self.gate = gate
        self.args = moe_args

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """inputs [T, D] (already ffn-normalised) -> sum over the token's k experts of w * expert(inputs).

        Router = one kernel (bf16 logits, top-k, fp32 softmax over the picks); the (token, slot) pairs are then grouped
        by expert with one stable sort, each expert that received tokens runs its fused SwiGLU FFN on exactly those rows,
        and the weighted outputs are added expert by expert in ascending id - the order in which the reference's loop
        (moe.py:29-31) rounds its bf16 accumulator."""
        k = self.args.num_experts_per_tok
        idx, w = _hip.moe_router(inputs, self.gate.weight, k)          # [T, k] int32 / fp32 (bf16-valued)
        flat_e = idx.flatten().long()
        order = torch.argsort(flat_e, stable=True)                      # pairs grouped by expert, token order kept
        counts = torch.bincount(flat_e, minlength=len(self.experts)).tolist()
        tok_of = torch.div(order, k, rounding_mode="floor")
        gate_w = w.flatten()[order].to(inputs.dtype)
        out = torch.zeros_like(inputs)
        start = 0
        for e, n in enumerate(counts):
            if n:
                rows = tok_of[start:start + n]
                y = self.experts[e](inputs.index_select(0, rows))
                out.index_add_(0, rows, gate_w[start:start + n, None] * y)  # a token meets an expert at most once
                start += n
        return out
