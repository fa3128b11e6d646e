"""Loading of tests/golden/vision_* (outputs of the unmodified reference's Pixtral path, oracle/make_golden_vision.py)."""
import json
import os

import torch
from safetensors.torch import load_file

import mistral_oracle as mo
import vision_oracle as vo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "vision_index.json")) as f:
    VINDEX = json.load(f)
VCASES = sorted(VINDEX)


class VisionCase:
    def __init__(self, name: str):
        self.name = name
        self.meta = VINDEX[name]
        self.t = load_file(os.path.join(GOLDEN, f"{name}.safetensors"))
        self.params = self.meta["params"]
        self.dtype = getattr(torch, self.meta["dtype"])
        self.text_args = mo.OracleArgs.from_params({k: v for k, v in self.params.items() if k != "vision_encoder"})
        self.vargs = vo.VisionArgs.from_params(self.params["vision_encoder"])
        self.prompt = torch.tensor(self.meta["prompt"], dtype=torch.long)
        self.images = [self.t[f"image.{i}"].to(self.dtype) for i in range(len(self.meta["sizes"]))]

    def weights(self):
        w = {k: v.to(self.dtype) for k, v in mo.synth_weights(self.text_args, seed=self.meta["text_seed"]).items()}
        w.update({k: v.to(self.dtype) for k, v in
                  vo.synth_vision_weights(self.vargs, self.text_args.dim, seed=self.meta["vision_seed"]).items()})
        chk = float(sum(v.double().abs().sum().item() for v in w.values()))
        assert chk == self.meta["weights_checksum"], "synthetic weights no longer regenerate bit-identically"
        return w
