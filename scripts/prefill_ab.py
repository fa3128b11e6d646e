#!/usr/bin/env python3
"""Same-process, same-box A/B of the prefill kernels' alternative forms (mi_debug_set_prefill_kernels): the causal attention of
a Mistral-7B layer at 4096 tokens in 256-query (8-wave) / 128-query (4-wave) blocks, and the q|k|v GEMM (16 x 24 square tiles = 1.5 rounds) with its tail on the 128-tile kernel / as half-height tiles / unsplit.

    gpurun --timeout 600 -- 'python scripts/prefill_ab.py [reps]'

Per form: microseconds (median over `reps` interleaved passes of 5 x 10 launches, HIP events on the launch stream), TFLOP/s and
a digest of the output bits (all forms of one op must agree, except the tail on the 128-tile kernel, which may differ by the
operand order of its MFMAs)."""
import hashlib
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))


def main():
    import torch
    from mistral_inference import _hip

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev, dtype=torch.float32) * scale).to(torch.bfloat16)

    def shapes(T, D, H, KV, DH=128):
        x = rnd(T, D)
        wq, wk, wv = rnd(H * DH, D, scale=0.02), rnd(KV * DH, D, scale=0.02), rnd(KV * DH, D, scale=0.02)
        qkv_in = rnd(T, (H + 2 * KV) * DH)
        q_start = torch.tensor([0, T], dtype=torch.int32, device=dev)
        kv_before = torch.tensor([0], dtype=torch.int32, device=dev)
        return dict(
            qkv=(lambda: _hip.linear(x, (wq, wk, wv), _hip.EPI_STORE), 2.0 * T * D * (H + 2 * KV) * DH),
            attn=(lambda: _hip.attn_prefill(qkv_in, H, KV, DH, None, None, T, q_start, kv_before, 1, T), 4.0 * T * T * H * DH / 2))

    cfgs = {"7b_4096": shapes(4096, 4096, 32, 8), "nemo_8192": shapes(8192, 5120, 32, 8)}
    forms = {
        "attn": [("blocks_of_256_queries", dict(attn_waves=8)), ("blocks_of_128_queries", dict(attn_waves=4))],
        "qkv": [("tail_128_kernel", dict(gemm_tail=1)), ("tail_half_height", dict(gemm_tail=2)), ("unsplit", dict(gemm_tail=0))],
    }
    res = {}
    for rep in range(reps):
        for cname, ops in cfgs.items():
            for op, (fn, flops) in ops.items():
                for fname, kw in forms[op]:
                    _hip.debug_set_prefill_kernels(**kw)
                    y = fn()
                    torch.cuda.synchronize()
                    key = f"{cname}.{op}.{fname}"
                    r = res.setdefault(key, {"us": [], "flops": flops})
                    if rep == 0:
                        r["sha"] = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
                        r["finite"] = bool(torch.isfinite(y.float()).all())
                    for _ in range(3):
                        fn()
                    ts = []
                    for _ in range(5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(10):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) * 100.0)
                    ts.sort()
                    r["us"].append(round(ts[2], 1))
    _hip.debug_set_prefill_kernels(attn_waves=0, gemm_tail=2)
    for key, r in res.items():
        us = sorted(r["us"])[len(r["us"]) // 2]
        print(f"{key:42s} {us:8.1f} us  {r['flops'] / us / 1e6:7.1f} TFLOP/s  sha {r['sha']}  finite {r['finite']}  all {r['us']}", flush=True)
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
