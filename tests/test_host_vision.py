"""Host-side (pure reshaping) pieces of the vision path against the reference-pinned oracle, on CPU."""
import torch

import vision_oracle as vo
from mistral_inference.vision_encoder import PatchMerger, position_meshgrid, precompute_freqs_cis_2d


def test_rope_2d_table_and_positions():
    ref = vo.rope_cs_2d(64, 5, 5, 10000.0)
    got = torch.view_as_real(precompute_freqs_cis_2d(64, 5, 5, 10000.0))
    assert torch.equal(got, ref)
    pos = position_meshgrid([(2, 3), (1, 2)])
    assert pos.tolist() == [[0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [1, 2], [0, 0], [0, 1]]


def test_patch_merger_permutation_is_unfold_order():
    g = torch.Generator().manual_seed(0)
    grids, d, s = [(4, 2), (2, 6)], 8, 2
    x = torch.randn(sum(h * w for h, w in grids), d, generator=g)
    eye = torch.eye(d * s * s)  # identity merging layer: patch_merge returns the permuted rows themselves
    ref = vo.patch_merge(x, grids, s, eye)
    with torch.device("meta"):
        pm = PatchMerger(d, s)
    assert torch.equal(pm.permute(x, grids), ref)
    # and the same thing the reference computes with F.unfold
    rows, o = [], 0
    for gh, gw in grids:
        grid = x[o:o + gh * gw].view(gh, gw, d).permute(2, 0, 1)[None]
        u = torch.nn.functional.unfold(grid, kernel_size=s, stride=s).view(1, d, s, s, -1)[0]
        rows.append(u.reshape(-1, u.shape[-1]).t())
        o += gh * gw
    assert torch.equal(torch.cat(rows), ref)
