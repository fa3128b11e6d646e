"""Seeded logits rows of the nucleus-sampling goldens (shared by oracle/make_golden_sampling.py and the tests: the rows are
regenerated from the seed, only the reference's OUTPUTS are stored)."""
import torch


def sampling_cases():
    """name -> (logits row fp32 [V], temperature, top_p); shared with the tests (they regenerate the rows from the seeds)."""
    g = torch.Generator().manual_seed(2024)
    cases = {}
    flat = (torch.randn(512, generator=g) * 0.58).to(torch.bfloat16).float()           # a random-init model's logits (bf16-valued)
    peaked = (torch.randn(512, generator=g) * 4.0).to(torch.bfloat16).float()          # a trained model's: few tokens carry the mass
    ties = torch.tensor([2.0, 1.0, 2.0, 1.0, 0.5, 2.0, 1.0, -3.0] * 16)                # exact ties everywhere (bf16 logits tie a lot)
    wide = torch.randn(1000, generator=g) * 2.5                                        # arbitrary fp32 values, odd vocabulary
    for nm, row in (("flat", flat), ("peaked", peaked), ("ties", ties), ("wide", wide)):
        for t in (0.7, 1.0, 0.3):
            for p in (0.8, 0.5):
                cases[f"{nm}_t{t}_p{p}"] = (row, t, p)
    cases["flat_t0.7_p1.0"] = (flat, 0.7, 1.0)
    cases["peaked_t0.7_p0.0"] = (peaked, 0.7, 0.0)
    return cases
