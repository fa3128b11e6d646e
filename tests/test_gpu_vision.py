"""Pixtral vision path on the GPU against outputs of the unmodified reference
(tests/golden/vision_*_bf16) and against the CPU oracle: tower output, merged embeddings, logits of the multimodal
forward, generate() with images; plus the two leaf additions it needed (mi_gelu, explicit softmax scale)."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

import mistral_oracle as mo
import vision_oracle as vo
from vision_util import VCASES, VisionCase

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
BF_CASES = [c for c in VCASES if c.endswith("bf16")]


def _load(tmp_path, case, max_batch_size=2):
    from mistral_inference.transformer import Transformer
    folder = tmp_path / "ckpt"
    os.makedirs(folder, exist_ok=True)
    with open(folder / "params.json", "w") as f:
        json.dump(case.params, f)
    save_file({k: v.contiguous() for k, v in case.weights().items()}, str(folder / "consolidated.safetensors"))
    return Transformer.from_folder(folder, max_batch_size=max_batch_size, device="cuda", dtype=BF)


def _close(got, ref, rel):
    return float((got.float().cpu() - ref.float()).abs().max()) <= rel * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("name", BF_CASES)
def test_vision_tower_embeddings_and_logits(name, tmp_path):
    c = VisionCase(name)
    model = _load(tmp_path, c)
    imgs = [im.cuda() for im in c.images]
    enc = model.vision_encoder(imgs)
    assert enc.shape == c.t["encoder_out"].shape
    assert _close(enc, c.t["encoder_out"], 4e-2), float((enc.float().cpu() - c.t["encoder_out"]).abs().max())
    w = c.weights()
    assert _close(enc, vo.vision_encoder(c.images, w, c.vargs), 4e-2)
    emb = model.embed_vision_language_features(c.prompt.cuda(), imgs)
    assert _close(emb, c.t["embeddings"], 4e-2)
    T = c.prompt.numel()
    logits = model.forward(c.prompt.cuda(), [T], images=imgs)
    assert logits.dtype == torch.float32
    assert float((logits.cpu() - c.t["logits"]).abs().max()) <= 6e-2
    # text-only forward on the same model still goes through the embedding kernel
    text = torch.tensor([1, 20, 21, 22, 27], dtype=torch.long)
    ref = mo.OracleModel(c.text_args, w).forward(text, [5], None)
    assert float((model.forward(text.cuda(), [5]).cpu() - ref).abs().max()) <= 4e-2


@pytest.mark.parametrize("name", [c for c in VCASES if c.endswith("fp32")])
def test_vision_tower_embeddings_and_logits_fp32(name, tmp_path):
    """The same three checkpoints of the Pixtral path in fp32 storage, on the generic kernels (64-wide heads as they are, no
    zero padding): tower output, merged embeddings and multimodal logits against the UNMODIFIED reference's stored fp32 outputs."""
    from mistral_inference.transformer import Transformer
    c = VisionCase(name)
    folder = tmp_path / "ckpt"
    os.makedirs(folder, exist_ok=True)
    with open(folder / "params.json", "w") as f:
        json.dump(c.params, f)
    save_file({k: v.contiguous() for k, v in c.weights().items()}, str(folder / "consolidated.safetensors"))
    model = Transformer.from_folder(folder, max_batch_size=2, device="cuda", dtype=torch.float32)
    imgs = [im.cuda() for im in c.images]
    enc = model.vision_encoder(imgs)
    assert enc.dtype == torch.float32 and enc.shape == c.t["encoder_out"].shape
    e1 = float((enc.cpu() - c.t["encoder_out"]).abs().max())
    emb = model.embed_vision_language_features(c.prompt.cuda(), imgs)
    e2 = float((emb.cpu() - c.t["embeddings"]).abs().max())
    T = c.prompt.numel()
    logits = model.forward(c.prompt.cuda(), [T], images=imgs)
    e3 = float((logits.cpu() - c.t["logits"]).abs().max())
    print(f"\n{name}: max |HIP - reference|: tower {e1:.3e}, embeddings {e2:.3e}, logits {e3:.3e}")
    scale = max(1.0, float(c.t["encoder_out"].abs().max()))
    assert e1 <= 2e-5 * scale and e2 <= 2e-5 * max(1.0, float(c.t["embeddings"].abs().max())) and e3 <= 2e-5 * max(1.0, float(c.t["logits"].abs().max()))


def test_generate_with_images(tmp_path):
    from mistral_inference.generate import generate
    c = VisionCase("vision_pixtral_bf16")
    model = _load(tmp_path, c)
    toks, lps = generate([c.prompt.tolist()], model, images=[[im.float().numpy() for im in c.images]], max_tokens=4,
                         temperature=0.0)
    assert len(toks) == 1 and len(toks[0]) == 4 and len(lps[0]) == c.prompt.numel() - 1 + 4
    # first generated token = argmax of the reference's last-row logits (unless the top two are within bf16 noise)
    last = c.t["logits"][-1]
    top2 = torch.topk(last, 2).values
    if float(top2[0] - top2[1]) > 8e-2:
        assert toks[0][0] == int(last.argmax())
    with pytest.raises(AssertionError):  # the reference refuses chunked prefill with images (generate.py:56)
        generate([c.prompt.tolist()], model, images=[[im.float().numpy() for im in c.images]], max_tokens=1,
                 temperature=0.0, chunk_size=4)


def test_gelu_and_softmax_scale_ops():
    from mistral_inference import _hip as h
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 200, generator=g) * 2).to(BF)
    got = h.gelu_(x.clone().cuda()).cpu()
    ref = torch.nn.functional.gelu(x.float()).to(BF)
    assert float((got.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    # 64-wide heads run zero-padded to 128 with scale 64^-1/2: must equal plain attention on the 64 real columns
    T, H = 50, 2
    qkv64 = torch.randn(T, 3 * H, 64, generator=g).to(BF)
    pad = torch.zeros(T, 3 * H, 128, dtype=BF)
    pad[:, :, :64] = qkv64
    out = h.attn_prefill(pad.view(T, -1).cuda(), H, H, 128, None, None, T, None, None, 1, T, causal=False,
                         softmax_scale=64 ** -0.5).cpu().view(T, H, 128)
    q, k, v = (qkv64[:, i * H:(i + 1) * H].float().transpose(0, 1) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(1, 2) * 64 ** -0.5, -1) @ v).transpose(0, 1)
    assert float((out[:, :, :64].float() - ref).abs().max()) <= 2.5e-2
    assert float(out[:, :, 64:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [BF, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("merger", [False, True])
def test_reference_selfconsistency_with_images(merger, dtype):
    """The reference's own two Pixtral tests (tests/test_generate.py:72-171) on the HIP path: generate 7 tokens with
    images, then re-score prompt + generation in ONE prefill; every logprob must agree.  fp32 storage (what the reference's
    tests use, on the generic kernels): the reference's own bound, 5e-4.  bf16 storage: 1 ulp at 1.0 is 7.8e-3.  Same tiny
    shapes as the reference, including 2-pixel patches (C*P*P = 12: the patch GEMM's K is zero-padded to 16)."""
    import numpy as np
    from mistral_inference.args import TransformerArgs, VisionEncoderArgs
    from mistral_inference.generate import generate
    from mistral_inference.transformer import Transformer
    torch.manual_seed(42)
    gen = np.random.default_rng(seed=42)
    side = 8 if merger else 4
    seqs = [[1, 2, 2, 2, 2, 4, 5, 6, 7], [12, 13, 14], [2, 2, 2, 2, 7, 8, 9]]
    images = [[gen.normal(size=(3, side, side))], [], [gen.normal(size=(3, side, side))]]
    extra = dict(adapter_bias=False, spatial_merge_size=2, add_pre_mm_projector_layer_norm=True,
                 mm_projector_id="patch_merge") if merger else {}
    args = TransformerArgs(dim=512, n_layers=1, head_dim=128, hidden_dim=2048, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
                           vocab_size=32_000, max_batch_size=len(seqs),
                           vision_encoder=VisionEncoderArgs(hidden_size=128, num_channels=3, image_size=side, patch_size=2,
                                                            intermediate_size=256, num_hidden_layers=1,
                                                            num_attention_heads=2, rope_theta=10000, image_token_id=2,
                                                            **extra))
    model = Transformer(args).to("cuda", dtype=dtype)
    toks, lp_old = generate(seqs, model, images=images, temperature=0.0, max_tokens=7)
    enc2 = [e + t for e, t in zip(seqs, toks)]
    generated, lp_new = generate(enc2, model, images=images, temperature=0.0, max_tokens=0)
    assert generated == []
    assert len(seqs) == len(lp_old) == len(lp_new)
    worst = max(abs(x - y) for a, b in zip(lp_old, lp_new) for x, y in zip(a, b))
    assert worst < (8e-2 if dtype == BF else 5e-4), worst
