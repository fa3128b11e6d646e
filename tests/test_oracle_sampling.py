"""The oracle's restatement of the reference's nucleus sampler (oracle/mistral_oracle.py::top_p_distribution,
generate.py:151-167) against what the UNMODIFIED reference handed to torch.multinomial on seeded logits rows
(tests/golden/sampling.safetensors, oracle/make_golden_sampling.py)."""
import json
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mistral_oracle as mo  # noqa: E402
from make_golden_sampling_cases import sampling_cases  # noqa: E402

GOLD = load_file(os.path.join(ROOT, "tests", "golden", "sampling.safetensors"))
INDEX = json.load(open(os.path.join(ROOT, "tests", "golden", "sampling_index.json")))["cases"]


@pytest.mark.parametrize("name", sorted(INDEX))
def test_top_p_distribution_matches_the_reference(name):
    row, t, p = sampling_cases()[name]
    assert (t, p) == (INDEX[name]["temperature"], INDEX[name]["top_p"])
    kept_ref, order_ref = GOLD[name + ".kept_sorted"].double(), GOLD[name + ".order"].long()
    order, kept = mo.top_p_distribution(row, t, p)
    n = INDEX[name]["n_kept"]
    assert int((kept > 0).sum()) == n                       # the same number of tokens survives the cut
    # the reference's torch.sort is free to order exact ties either way; per-TOKEN probabilities must agree
    ref_tok = torch.zeros(row.numel(), dtype=torch.float64).scatter_(0, order_ref, kept_ref)
    got_tok = torch.zeros(row.numel(), dtype=torch.float64).scatter_(0, order, kept)
    probs = torch.softmax(row / t, dim=-1)
    tied_at_cut = probs == probs[order[n - 1]]              # which of the tokens tied at the boundary value survive is the sort's choice
    free = ~tied_at_cut
    assert float((ref_tok[free] - got_tok[free]).abs().max()) <= 1e-6
    assert abs(float(ref_tok[tied_at_cut].sum() - got_tok[tied_at_cut].sum())) <= 1e-6
    assert torch.equal(kept_ref[:n] > 0, kept[:n] > 0) and float(kept.sum()) == pytest.approx(1.0, abs=1e-12)


def test_inverse_cdf_draw_walks_the_kept_prefix():
    row, t, p = sampling_cases()["peaked_t0.7_p0.8"]
    order, kept = mo.top_p_distribution(row, t, p)
    n = int((kept > 0).sum())
    assert mo.top_p_inverse_cdf(order, kept, 0.0) == int(order[0])
    assert mo.top_p_inverse_cdf(order, kept, 0.999999999) == int(order[n - 1])
    cdf = torch.cumsum(kept, 0)
    for k in range(n):
        u = float(cdf[k]) - 0.5 * float(kept[k])
        assert mo.top_p_inverse_cdf(order, kept, u) == int(order[k])
