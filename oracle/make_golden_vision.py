#!/usr/bin/env python3
"""Generate tests/golden/vision_*.safetensors by running the UNMODIFIED reference (Pixtral vision path) on CPU.

    python oracle/make_golden_vision.py        # build container only (needs /root/reference)

Tiny multimodal models (text: 2 layers, dim 256; vision tower: 2 layers, 2 heads of 64, 16-pixel patches) with weights from
`mistral_oracle.synth_weights` + `vision_oracle.synth_vision_weights` (regenerated in the tests, checksum stored).  Stored
tensors are OUTPUTS of the reference: `vision_encoder(images)`, `embed_vision_language_features`, `forward` logits.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")
sys.path[:0] = [os.path.join(HERE, "shim"), REF, HERE]

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import mistral_oracle as mo  # noqa: E402
import vision_oracle as vo  # noqa: E402
from mistral_inference.args import TransformerArgs  # noqa: E402  (the reference)
from mistral_inference.transformer import Transformer  # noqa: E402

assert os.path.realpath(sys.modules["mistral_inference"].__file__).startswith(os.path.realpath(REF))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TEXT = dict(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=512)
VISION = dict(hidden_size=128, num_channels=3, image_size=64, patch_size=16, intermediate_size=256, num_hidden_layers=2,
              num_attention_heads=2, rope_theta=10000.0, image_token_id=10)
CASES = {
    # name: (vision overrides, dtype, image sizes (H, W))
    "vision_pixtral_fp32": (dict(), "float32", [(32, 48), (64, 32)]),
    "vision_pixtral_bf16": (dict(), "bfloat16", [(32, 48), (64, 32)]),
    "vision_merge_fp32": (dict(adapter_bias=False, add_pre_mm_projector_layer_norm=True, mm_projector_id="patch_merge",
                               spatial_merge_size=2), "float32", [(64, 32), (32, 64)]),
    "vision_merge_bf16": (dict(adapter_bias=False, add_pre_mm_projector_layer_norm=True, mm_projector_id="patch_merge",
                               spatial_merge_size=2), "bfloat16", [(64, 32), (32, 64)]),
}


def prompt_for(n_img_tokens):
    ids, t = [1], 20
    for n in n_img_tokens:
        ids += [10] * n + [t, t + 1, t + 2]
        t += 7
    return ids


def main() -> None:
    index = {}
    for name, (over, dt, sizes) in CASES.items():
        dtype = getattr(torch, dt)
        vp = dict(VISION)
        vp.update(over)
        params = dict(TEXT)
        params["vision_encoder"] = vp
        va = vo.VisionArgs.from_params(vp)
        w = {k: v.to(dtype) for k, v in mo.synth_weights(mo.OracleArgs.from_params(TEXT), seed=42).items()}
        w.update({k: v.to(dtype) for k, v in vo.synth_vision_weights(va, TEXT["dim"], seed=43).items()})
        rargs = TransformerArgs.from_dict(params)
        rargs.max_batch_size = 2
        model = Transformer(rargs)
        model.load_state_dict({k: v.clone() for k, v in w.items()}, assign=True, strict=True)
        model = model.to("cpu", dtype=dtype).eval()
        g = torch.Generator().manual_seed(5)
        images = [torch.randn(3, h, wd, generator=g).to(dtype) for h, wd in sizes]
        s = va.spatial_merge_size if va.mm_projector_id == "patch_merge" else 1
        n_tok = [(h // 16 // s) * (wd // 16 // s) for h, wd in sizes]
        ids = torch.tensor(prompt_for(n_tok), dtype=torch.long)
        with torch.no_grad():
            enc = model.vision_encoder(images)
            emb = model.embed_vision_language_features(ids, images)
            logits = model.forward(ids, [ids.numel()], images=images)
        tens = {"encoder_out": enc.float(), "embeddings": emb.float(), "logits": logits.float()}
        for i, im in enumerate(images):
            tens[f"image.{i}"] = im.float()
        save_file({k: v.contiguous() for k, v in tens.items()}, os.path.join(OUT, f"{name}.safetensors"))
        index[name] = dict(params=params, dtype=dt, sizes=sizes, prompt=ids.tolist(), text_seed=42, vision_seed=43,
                           weights_checksum=float(sum(v.double().abs().sum().item() for v in w.values())))
        print(name, "encoder", tuple(enc.shape), "logits", tuple(logits.shape), float(logits.abs().max()))
    with open(os.path.join(OUT, "vision_index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
