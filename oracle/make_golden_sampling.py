#!/usr/bin/env python3
"""Golden vectors for the nucleus sampler: what the UNMODIFIED reference `sample()` (generate.py:151-170) hands to
torch.multinomial - the sorted, masked, renormalised probabilities - and the permutation torch.sort produced, on seeded
logits rows.  torch.multinomial is replaced by a recorder for the call (its draw is torch's own stream and pins nothing);
everything before it is the reference's code, untouched.

    python oracle/make_golden_sampling.py        # build container only (needs /root/reference)

Writes tests/golden/sampling.safetensors (+ the case list in sampling_index.json).  tests/test_oracle_sampling.py holds
oracle/mistral_oracle.py::top_p_distribution to these; the GPU kernel is then tested against the oracle."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")
sys.path[:0] = [os.path.join(HERE, "shim"), REF, HERE]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import mistral_inference.generate as ref_generate  # noqa: E402  (the reference)

assert os.path.realpath(ref_generate.__file__).startswith(os.path.realpath(REF)), "golden vectors must come from the reference package"

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


from make_golden_sampling_cases import sampling_cases  # noqa: E402


def main():
    rec = {}
    real = torch.multinomial

    def recorder(probs, num_samples, *a, **k):
        rec["kept"] = probs.clone()
        return real(probs, num_samples, *a, **k)

    tensors, index = {}, {}
    for name, (row, t, p) in sampling_cases().items():
        real_sort = torch.sort

        def sort_rec(x, *a, **k):
            out = real_sort(x, *a, **k)
            rec["order"] = out[1].clone()
            return out

        torch.multinomial, torch.sort = recorder, sort_rec
        try:
            tok = ref_generate.sample(row[None, :].clone(), temperature=t, top_p=p)
        finally:
            torch.multinomial, torch.sort = real, real_sort
        kept, order = rec["kept"][0], rec["order"][0]
        assert int(tok) in set(order[kept > 0].tolist())
        tensors[name + ".kept_sorted"] = kept.contiguous()          # fp32 [V]: the distribution multinomial drew from
        tensors[name + ".order"] = order.to(torch.int32).contiguous()
        index[name] = {"temperature": t, "top_p": p, "vocab": int(row.numel()), "n_kept": int((kept > 0).sum())}
    os.makedirs(OUT, exist_ok=True)
    save_file(tensors, os.path.join(OUT, "sampling.safetensors"))
    with open(os.path.join(OUT, "sampling_index.json"), "w") as f:
        json.dump({"torch": torch.__version__, "cases": index}, f, indent=1)
    print(f"wrote {len(index)} sampling cases")


if __name__ == "__main__":
    main()
