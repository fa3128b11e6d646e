"""Loading of tests/golden (produced by oracle/make_golden.py from the unmodified reference)."""
import json
import os

import torch
from safetensors.torch import load_file

import mistral_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(GOLDEN, "index.json")) as f:
    INDEX = json.load(f)

CASES = sorted(INDEX)
# cases replayed on the GPU; `oracle_only` entries pin the CPU oracle to more reference outputs without being part of it
GPU_CASES = [c for c in CASES if not INDEX[c].get("oracle_only") and INDEX[c]["dtype"] != "float16"]
# cases replayed on the GPU in their OWN storage dtype through mi_forward_generic (tests/test_gpu_generic.py)
GENERIC_CASES = [c for c in CASES if INDEX[c]["dtype"] in ("float32", "float16")]


class Case:
    def __init__(self, name: str):
        self.name = name
        self.meta = INDEX[name]
        self.t = load_file(os.path.join(GOLDEN, f"{name}.safetensors"))
        self.params = self.meta["params"]
        self.dtype = getattr(torch, self.meta["dtype"])
        self.args = mo.OracleArgs.from_params(self.params)
        self.prompts = self.meta["prompts"]
        self.max_tokens = self.meta["max_tokens"]
        self.chunk_size = self.meta["chunk_size"]
        self.max_batch_size = self.meta["max_batch_size"]

    def weights(self):
        w = mo.synth_weights(self.args, seed=self.meta["seed"], dtype=torch.bfloat16)
        w = {k: v.to(self.dtype) for k, v in w.items()}
        chk = float(sum(v.double().abs().sum().item() for v in w.values()))
        assert chk == self.meta["weights_checksum"], "synthetic weights no longer regenerate bit-identically"
        return w

    def logprobs(self):
        lp = self.t["logprobs"]
        return [[x for x in row.tolist() if x == x] for row in lp]

    def tokens(self):
        return self.t["tokens"].tolist()

    def n_prefill(self):
        return sum(1 for k in self.t if k.startswith("prefill_logits."))

    def n_decode(self):
        return sum(1 for k in self.t if k.startswith("decode_logits."))

    def tol(self):
        """(logit_atol, logprob_atol).  fp32: accumulation-order noise only.  bf16: the reference's own
        bf16 noise floor measured in SURVEY.md section 6 (7.8e-3 thread-count noise on 2 layers)."""
        if self.dtype == torch.float16:  # 10 mantissa bits against bf16's 7
            return (6e-3, 6e-3)
        return (2e-5, 2e-5) if self.dtype == torch.float32 else (4e-2, 4e-2)
