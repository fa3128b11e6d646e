"""Helpers for the -m gpu parity tests: synthetic checkpoints on disk + comparison utilities."""
import json
import os

import torch
from safetensors.torch import save_file

import mistral_oracle as mo


def write_checkpoint(folder, args: mo.OracleArgs, weights) -> str:
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "params.json"), "w") as f:
        json.dump(mo.params_json(args), f)
    save_file({k: v.contiguous() for k, v in weights.items()}, os.path.join(folder, "consolidated.safetensors"))
    return str(folder)


def bf16_ulp_close(got: torch.Tensor, ref: torch.Tensor, ulps: float = 1.0, floor: float = 1e-3):
    """|got - ref| <= ulps * (bf16 spacing at |ref|) (+ floor).  Returns (ok, worst absolute error)."""
    g, r = got.float().cpu(), ref.float().cpu()
    spacing = torch.clamp(r.abs(), min=floor) * 2.0 ** -7
    err = (g - r).abs()
    return bool((err <= ulps * spacing + 1e-6).all()), float(err.max())
