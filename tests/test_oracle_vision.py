"""Pins oracle/vision_oracle.py (the CPU restatement of the Pixtral vision path) to outputs of the unmodified reference."""
import pytest
import torch

import mistral_oracle as mo
import vision_oracle as vo
from vision_util import VCASES, VisionCase


@pytest.mark.parametrize("name", VCASES)
def test_vision_oracle_matches_reference(name):
    c = VisionCase(name)
    w = c.weights()
    tol = 3e-5 if c.dtype == torch.float32 else 4e-2
    enc = vo.vision_encoder(c.images, w, c.vargs)
    assert enc.shape == c.t["encoder_out"].shape
    assert float((enc.float() - c.t["encoder_out"]).abs().max()) <= tol * max(1.0, float(c.t["encoder_out"].abs().max()))
    emb = vo.embed_vision_language_features(c.prompt, c.images, w, c.vargs)
    assert float((emb.float() - c.t["embeddings"]).abs().max()) <= tol * max(1.0, float(c.t["embeddings"].abs().max()))
    model = mo.OracleModel(c.text_args, w)
    logits = model.forward(c.prompt, [c.prompt.numel()], None, h_in=emb)
    assert float((logits - c.t["logits"]).abs().max()) <= tol * 2


def test_patchify_is_conv2d():
    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 32, 48, generator=g)
    wt = torch.randn(8, 3, 16, 16, generator=g)
    ref = torch.nn.functional.conv2d(img[None], wt, stride=16)[0].flatten(1).permute(1, 0)
    got = torch.nn.functional.linear(vo.patchify(img, 16), wt.view(8, -1))
    assert torch.allclose(got, ref, atol=1e-4)
