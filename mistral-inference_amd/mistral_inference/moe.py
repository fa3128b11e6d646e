"""Sparse MoE layer (reference moe.py:10-32) on the gfx950 kernels.

Module-level forward (used when a block is driven layer by layer); `Transformer.forward_partial` runs the
same kernels from the native layer-stack runner.  At decode sizes (<= 8 tokens) the router, the selected
experts' gate/up GEMVs and the down-projection + weighted bf16 combine are three launches; larger batches
take the device-side expert sort + token-grouped GEMMs.  No torch compute and no host sync either way; the
reference needs `num_experts` `torch.where` syncs per layer (moe.py:30)."""
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

    def _expert_table(self) -> torch.Tensor:
        """int64 device tensor [E, 3] of the experts' (w1, w2, w3) weight pointers, rebuilt when a weight moved."""
        ptrs = [p for ex in self.experts for p in (ex.w1.weight.data_ptr(), ex.w2.weight.data_ptr(), ex.w3.weight.data_ptr())]
        if getattr(self, "_tab_ptrs", None) != ptrs:
            self._tab_ptrs = ptrs
            self._tab = torch.tensor(ptrs, dtype=torch.int64, device=self.gate.weight.device).view(-1, 3)
        return self._tab

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """inputs [T, D] (already ffn-normalised) -> sum over the token's k experts of w * expert(inputs), accumulated in
        bf16 in ascending expert id (the order in which the reference's loop rounds, moe.py:29-31).

        All device work is libmistral_hip: the router kernel (bf16 logits, top-k, fp32 softmax over the picks), then for
        T <= 8 the selected experts' fused gate/up GEMV + the down-projection/combine kernel (`mi_moe_experts_decode`),
        for larger T the on-device expert sort + ONE token-grouped MFMA GEMM launch per projection + ordered combine
        (`mi_moe_grouped_gemm`).  No host synchronisation (the reference does one `torch.where` sync per expert)."""
        k, E = self.args.num_experts_per_tok, len(self.experts)
        F, T = self.experts[0].w1.weight.shape[0], inputs.shape[0]
        # shapes the fused kernels are built for (csrc/api.hip check_model / mi_moe_grouped_gemm); the reference accepts any
        # num_experts / top_k, so everything else takes the general route below instead of raising
        fused = E <= 16 and k in (1, 2, 4) and k <= E and (T > _hip.GEMV_MAX_T or k * F * 2 <= 65536)
        if not fused:
            return self._forward_general(inputs)
        idx, w = _hip.moe_router(inputs, self.gate.weight, k)          # [T, k] int32 / fp32 (bf16-valued)
        return _hip.moe_experts(inputs, self._expert_table(), E, F, idx, w)

    def _forward_general(self, inputs: torch.Tensor) -> torch.Tensor:
        """Any expert count / top-k (reference moe.py:24-32 literally): the routing bookkeeping in torch on the device, the
        dense work - gate logits and every expert's SwiGLU FFN on its tokens - on the HIP GEMM / GEMV kernels (`mi_linear`).
        One `torch.where` host sync per expert, as in the reference."""
        k = self.args.num_experts_per_tok
        gate_logits = _hip.linear(inputs, (self.gate.weight,), _hip.EPI_STORE)
        weights, selected = torch.topk(gate_logits, k)
        weights = torch.softmax(weights, dim=1, dtype=torch.float).to(inputs.dtype)
        results = torch.zeros_like(inputs)
        for e, expert in enumerate(self.experts):
            tok, slot = torch.where(selected == e)
            if tok.numel() == 0:
                continue
            x = inputs.index_select(0, tok).contiguous()
            hid = _hip.linear(x, (expert.w1.weight, expert.w3.weight), _hip.EPI_SWIGLU)
            y = _hip.linear(hid, (expert.w2.weight,), _hip.EPI_STORE)
            results[tok] += weights[tok, slot, None] * y
        return results
