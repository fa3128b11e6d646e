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
        """inputs [T, D] (already ffn-normalised) -> sum_k w_k * expert_k(inputs) in the reference's bf16
        accumulation order.  Built from the leaf operators: router kernel, then per expert (ascending id,
        moe.py:29) the fused SwiGLU FFN on the tokens routed to it."""
        k = self.args.num_experts_per_tok
        idx, w = _hip.moe_router(inputs, self.gate.weight, k)
        results = torch.zeros_like(inputs)
        for e, expert in enumerate(self.experts):
            tok, slot = torch.where(idx == e)
            if tok.numel() == 0:
                continue
            y = expert(inputs[tok].contiguous())
            results[tok] += w[tok, slot, None].to(inputs.dtype) * y
        return results
