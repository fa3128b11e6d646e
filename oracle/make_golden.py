#!/usr/bin/env python3
"""Generate tests/golden/*.safetensors by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

The reference source (/root/reference/src/mistral_inference) is imported as-is; its two missing
third-party imports (xformers 0.0.26.post1, simple-parsing 0.1.5) come from oracle/shim.  Weights
are `mistral_oracle.synth_weights(args, seed)` (regenerated, not stored; a float64 checksum is stored
so a test can tell if regeneration ever drifts).  Every stored tensor is an OUTPUT of the reference:
`Transformer.forward` logits (hooked), per-layer block outputs (hooked) and `generate()` results.
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
from mistral_inference.args import TransformerArgs  # noqa: E402  (the reference)
from mistral_inference.generate import generate  # noqa: E402
from mistral_inference.transformer import Transformer  # noqa: E402

assert os.path.realpath(sys.modules["mistral_inference"].__file__).startswith(os.path.realpath(REF)), \
    "golden vectors must come from the reference package"

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY = dict(dim=256, n_layers=2, head_dim=128, hidden_dim=512, n_heads=4, n_kv_heads=2, norm_eps=1e-5,
            vocab_size=512)

CASES = {
    # name: (args overrides, dtype, prompts, max_tokens, chunk_size)
    "dense_fp32": (dict(), "float32", [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2], [11, 12, 13, 14, 15]], 6, None),
    "dense_bf16": (dict(), "bfloat16", [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2], [11, 12, 13, 14, 15]], 6, None),
    "swa_fp32": (dict(sliding_window=8), "float32",
                 [[(3 * i + 1) % 512 for i in range(13)], [5, 6, 7, 8, 9], [(7 * i + 2) % 512 for i in range(9)]], 12, None),
    "swa_bf16": (dict(sliding_window=8), "bfloat16",
                 [[(3 * i + 1) % 512 for i in range(13)], [5, 6, 7, 8, 9], [(7 * i + 2) % 512 for i in range(9)]], 12, None),
    "swa_chunk_fp32": (dict(sliding_window=8), "float32",
                       [[(3 * i + 1) % 512 for i in range(13)], [(7 * i + 2) % 512 for i in range(14)]], 5, 4),
    "swa_chunk_bf16": (dict(sliding_window=8), "bfloat16",
                       [[(3 * i + 1) % 512 for i in range(13)], [(7 * i + 2) % 512 for i in range(14)]], 5, 4),
    "swa_list_fp32": (dict(sliding_window=[4, None]), "float32",
                      [[(5 * i + 3) % 512 for i in range(10)], [9, 8, 7]], 7, None),
    "moe_fp32": (dict(moe=dict(num_experts=8, num_experts_per_tok=2)), "float32",
                 [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2]], 5, None),
    "moe_bf16": (dict(moe=dict(num_experts=8, num_experts_per_tok=2)), "bfloat16",
                 [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2]], 5, None),
    "rope_theta_fp32": (dict(rope_theta=10000.0), "float32", [[4, 3, 2, 1, 0, 9, 8, 7, 6, 5, 4, 3]], 4, None),
    # fp16 storage (from_folder(dtype=torch.float16), reference transformer.py:303,338): replayed on the GPU through
    # mi_forward_generic (tests/test_gpu_generic.py)
    "dense_fp16": (dict(), "float16", [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2], [11, 12, 13, 14, 15]], 6, None),
    "swa_chunk_fp16": (dict(sliding_window=8), "float16",
                       [[(3 * i + 1) % 512 for i in range(13)], [(7 * i + 2) % 512 for i in range(14)]], 5, 4),
    "moe_fp16": (dict(moe=dict(num_experts=8, num_experts_per_tok=2)), "float16",
                 [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2]], 5, None),
    # oracle-only pins (index flag `oracle_only`: not part of the GPU replay list): head layouts and option
    # combinations the cases above do not reach
    "mha_fp32": (dict(n_kv_heads=4), "float32", [[1, 5, 9, 200, 17, 3, 44], [7, 300, 2]], 5, None),
    "gqa8_swa_fp32": (dict(n_heads=8, n_kv_heads=1, sliding_window=6), "float32",
                      [[(11 * i + 5) % 512 for i in range(15)], [3, 1, 4, 1, 5, 9, 2, 6]], 9, None),
    "moe_swa_chunk_fp32": (dict(moe=dict(num_experts=8, num_experts_per_tok=2), sliding_window=8), "float32",
                           [[(3 * i + 1) % 512 for i in range(13)], [(7 * i + 2) % 512 for i in range(14)]], 5, 4),
}
ORACLE_ONLY = {"mha_fp32", "gqa8_swa_fp32", "moe_swa_chunk_fp32"}


def build(over, dtype):
    p = dict(TINY)
    p.update(over)
    oargs = mo.OracleArgs.from_params(p)
    w = mo.synth_weights(oargs, seed=42, dtype=torch.bfloat16)
    w = {k: v.to(dtype) for k, v in w.items()}
    rargs = TransformerArgs.from_dict(p)
    rargs.max_batch_size = 4
    model = Transformer(rargs)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, assign=True, strict=True)
    return p, oargs, w, model.to("cpu", dtype=dtype).eval()


def main(out_dir: str = OUT, only=None) -> None:
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    for name, (over, dt, prompts, max_tokens, chunk) in CASES.items():
        if only is not None and name not in only:
            continue
        dtype = getattr(torch, dt)
        params, oargs, w, model = build(over, dtype)
        fwd_out, layer_out = [], []
        hooks = []
        ref_forward = model.forward  # generate() calls model.forward(...) directly, so observe it by wrapping

        def observed_forward(*a, **k):
            o = ref_forward(*a, **k)
            fwd_out.append(o.detach().clone())
            return o

        model.forward = observed_forward
        for lid, layer in model.layers.items():
            hooks.append(layer.register_forward_hook(
                lambda m, i, o, lid=lid: layer_out.append((int(lid), o.detach().clone()))))
        with torch.inference_mode():
            toks, lps = generate(prompts, model, max_tokens=max_tokens, temperature=0.0, chunk_size=chunk)
        n_chunks = 1 if chunk is None else -(-max(len(p) for p in prompts) // chunk)
        tensors = {}
        for c in range(n_chunks):
            tensors[f"prefill_logits.{c}"] = fwd_out[c].float().contiguous()
        for s in range(n_chunks, len(fwd_out)):
            tensors[f"decode_logits.{s - n_chunks}"] = fwd_out[s].float().contiguous()
        # block outputs of the first forward only (one entry per layer)
        for lid, o in layer_out[: oargs.n_layers]:
            tensors[f"prefill_hidden.{lid}"] = o.float().contiguous()
        tensors["tokens"] = torch.tensor(toks, dtype=torch.int64)
        width = max(len(x) for x in lps)
        lp = torch.full((len(lps), width), float("nan"), dtype=torch.float64)
        for b, x in enumerate(lps):
            lp[b, : len(x)] = torch.tensor(x, dtype=torch.float64)
        tensors["logprobs"] = lp
        for h in hooks:
            h.remove()
        del model.forward

        # cache=None call (tutorials/classifier.ipynb cell 15 use-case; transformer_layers.py:165 quirk)
        if name.startswith("dense"):
            flat = torch.tensor(sum(prompts, []), dtype=torch.long)
            with torch.inference_mode():
                tensors["nocache_hidden"] = model.forward_partial(flat, [len(p) for p in prompts]).float().contiguous()

        meta = {
            "params": params, "dtype": dt, "prompts": prompts, "max_tokens": max_tokens, "chunk_size": chunk,
            "seed": 42, "max_batch_size": 4,
            "weights_checksum": float(sum(v.double().abs().sum().item() for v in w.values())),
        }
        if name in ORACLE_ONLY:
            meta["oracle_only"] = True
        save_file(tensors, os.path.join(out_dir, f"{name}.safetensors"))
        index[name] = meta
        print(f"{name}: {len(tensors)} tensors, tokens={toks}")
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
