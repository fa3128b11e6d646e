"""Prefill logits + K/V rings of two models with MI_FUSE_ROPE taken from the environment (the switch is read once per process):
tests/test_gpu_ops.py::test_fused_rope_epilogue_bit_equal_to_separate_pass runs this file twice and compares the dumps.

  small: 3 layers of toy dims, ragged 3-sequence batch of 200 tokens  -> every GEMM on the 128-tile kernel (M < 256)
  wide : 2 layers of the Mistral-7B dims, one 1000-token prompt then a 300-token chunk -> the 256-tile kernel with a ragged last
         m-tile (1000 = 3 x 256 + 232)
  tail : 1 layer of the same dims, 4096 tokens -> 1.5 rounds of square tiles: the k | v columns run as half-height tiles"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))

from bench import build_model  # noqa: E402
from mistral_inference.cache import BufferCache  # noqa: E402

CASES = {
    "small": (dict(dim=512, n_layers=3, head_dim=128, hidden_dim=1024, n_heads=4, n_kv_heads=2, norm_eps=1e-5, vocab_size=1024,
                   rope_theta=1e4), [[120, 47, 33]]),
    "wide": (dict(dim=4096, n_layers=2, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5, vocab_size=4096,
                  rope_theta=1e6), [[1000], [300]]),
    # 4096 tokens: q|k|v is 16 x 24 = 384 square tiles -> 256 of them + the k | v columns as 128 x 256 tiles (gemm256.hip, MH = 1)
    "tail": (dict(dim=4096, n_layers=1, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5, vocab_size=1024,
                  rope_theta=1e6), [[4096]]),
}


def main(out_path: str) -> None:
    dump = {}
    for name, (params, chunks) in CASES.items():
        m = build_model(params, 0, 1, "cuda")
        a = m.args
        B = len(chunks[0])
        a.max_batch_size = B  # (bench.build_model builds batch-1 models)
        total = [sum(c[b] for c in chunks) for b in range(B)]
        cache = BufferCache(m.n_local_layers, B, max(total), a.n_kv_heads, a.head_dim, None, device="cuda", dtype=torch.bfloat16)
        cache.reset()
        g = torch.Generator().manual_seed(7)
        for ci, lens in enumerate(chunks):
            ids = torch.randint(0, a.vocab_size, (sum(lens),), generator=g).cuda()
            dump[f"{name}.logits.{ci}"] = m.forward(ids, lens, cache).float().cpu()
        for l in range(m.n_local_layers):  # (only the written slots: the rings are torch.empty)
            for b in range(B):
                dump[f"{name}.k.{l}.{b}"] = cache.cache_k[l][b, :total[b]].float().cpu()
                dump[f"{name}.v.{l}.{b}"] = cache.cache_v[l][b, :total[b]].float().cpu()
        del m, cache
        torch.cuda.empty_cache()
    torch.save(dump, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
