#!/usr/bin/env python3
"""Same-box A/B of the prefill kernels (the four GEMMs of a Mistral-7B layer at 4096 tokens + causal attention) over
library variants: the main build and every mistral-inference_amd/lib/variants/libmistral_hip_*.so.

    python scripts/build_variants.py gemm
    gpurun --timeout 900 -- 'python scripts/prefill_probe.py [reps] [name-substring ...]'

Every library runs in its own process (the library is chosen at import: MISTRAL_HIP_LIB), `reps` passes interleaved
over the libraries (consecutive runs on one box drift by ~1.5 %).  Per op: microseconds (median of 5 x 10 launches,
HIP events on the launch stream), TFLOP/s, and whether the output bits equal the main build's (every schedule variant
keeps the accumulation order, so they must; the `abl_*` timing ablations are wrong by construction)."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "mistral-inference_amd")


def worker():
    sys.path.insert(0, PKG)
    import torch
    from mistral_inference import _hip

    dev = torch.device("cuda:0")
    T, D, F, H, KV, DH = 4096, 4096, 14336, 32, 8, 128
    torch.manual_seed(1234)  # (device generator: the same values in every process on one box)

    def rnd(*shape, scale=1.0):
        if os.environ.get("PROBE_ZERO"):  # zero operands: same instruction stream, (almost) no switching power
            return torch.zeros(*shape, device=dev, dtype=torch.bfloat16)
        return (torch.randn(*shape, device=dev, dtype=torch.float32) * scale).to(torch.bfloat16)

    x = rnd(T, D)
    res = rnd(T, D)
    hid = rnd(T, F, scale=0.5)
    wq, wk, wv = rnd(H * DH, D, scale=0.02), rnd(KV * DH, D, scale=0.02), rnd(KV * DH, D, scale=0.02)
    wo = rnd(D, H * DH, scale=0.02)
    w1, w3 = rnd(F, D, scale=0.02), rnd(F, D, scale=0.02)
    w2 = rnd(D, F, scale=0.02)
    qkv_in = rnd(T, (H + 2 * KV) * DH)
    q_start = torch.tensor([0, T], dtype=torch.int32, device=dev)
    kv_before = torch.tensor([0], dtype=torch.int32, device=dev)

    ops = {
        "qkv": (lambda: _hip.linear(x, (wq, wk, wv), _hip.EPI_STORE), 2.0 * T * D * (H + 2 * KV) * DH),
        "wo": (lambda: _hip.linear(x, (wo,), _hip.EPI_RESIDUAL, residual=res), 2.0 * T * D * D),
        "w13": (lambda: _hip.linear(x, (w1, w3), _hip.EPI_SWIGLU), 2.0 * T * D * 2 * F),
        "w2": (lambda: _hip.linear(hid, (w2,), _hip.EPI_RESIDUAL, residual=res), 2.0 * T * F * D),
        "attn": (lambda: _hip.attn_prefill(qkv_in, H, KV, DH, None, None, 4096, q_start, kv_before, 1, T), 4.0 * T * T * H * DH / 2),
    }
    if os.environ.get("PROBE_POWER"):  # board power / clocks while one op loops (rocm-smi sampled from a thread)
        import threading, time
        samples, stop = [], False

        def sampler():
            while not stop:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
                try:
                    c = json.loads(r.stdout)["card0"]
                    samples.append({k: v for k, v in c.items() if "ower" in k or "sclk" in k or "mclk" in k})
                except Exception as e:
                    samples.append({"error": str(e)[:80], "out": r.stdout[:200]})
                time.sleep(0.05)

        for name in os.environ["PROBE_POWER"].split(","):
            fn = ops[name][0]
            stop, samples = False, []
            th = threading.Thread(target=sampler)
            th.start()
            t0 = time.time()
            while time.time() - t0 < 3.0:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
            stop = True
            th.join()
            print("POWER " + name + " " + json.dumps(samples[len(samples) // 2:][:6]), flush=True)
        return
    out = {}
    if os.environ.get("PROBE_CHECK"):  # independent sanity of the library under test (fp32 torch matmul, bf16-rounded)
        ref = (x.float() @ torch.cat([wq, wk, wv]).float().T).to(torch.bfloat16).float()
        out["check_qkv_maxdiff"] = float((ops["qkv"][0]().float() - ref).abs().max())
        ref = ((hid.float() @ w2.float().T).to(torch.bfloat16).float() + res.float()).to(torch.bfloat16).float()
        out["check_w2_maxdiff"] = float((ops["w2"][0]().float() - ref).abs().max())
        print("CHECK " + json.dumps(out), flush=True)
        out = {}
    for name, (fn, flops) in ops.items():
        y = fn()
        torch.cuda.synchronize()
        digest = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        finite = bool(torch.isfinite(y.float()).all())
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
            ts.append(e0.elapsed_time(e1) * 100.0)  # us per launch
        ts.sort()
        clk = None
        if hasattr(_hip.lib(), "mi_debug_gemm_clock") and name != "attn":  # G256_CLK builds: ticks of block 0's main loop
            import ctypes
            buf = (ctypes.c_ulonglong * 2)()
            _hip.lib().mi_debug_gemm_clock(buf)
            clk = [int(buf[0]), int(buf[1])]
        out[name] = {"clk": clk, "us": round(ts[2], 1), "min_us": round(ts[0], 1), "tf": round(flops / ts[2] / 1e6, 1), "sha": digest, "finite": finite}
    print("RESULT " + json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--power":  # python scripts/prefill_probe.py --power w13,attn [zeros]
        env = dict(os.environ, PROBE_POWER=args[1] if len(args) > 1 else "w13")
        for zeros in ([False, True] if "zeros" in args else [False]):
            if zeros:
                env["PROBE_ZERO"] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, timeout=300)
            print(("zero operands: " if zeros else "random operands: ") + "\n".join(l for l in r.stdout.splitlines() if l.startswith("POWER")) + r.stderr[-300:], flush=True)
        return
    reps = int(args[0]) if args and args[0].isdigit() else 2
    filt = [a for a in args if not a.isdigit()]
    libs = [("main", None, {})]
    # run-time switches of the main build (A/B knobs that already exist)
    for n, e in (("env_nostagger", {"MI_GEMM_STAGGER": "0"}), ("env_notail", {"MI_GEMM_TAIL": "0"}),
                 ("env_attn_waves4", {"MI_ATTN_PREFILL_WAVES": "4"})):
        if not filt or any(s in n for s in filt):
            libs.append((n, None, e))
    for f in sorted(glob.glob(os.path.join(PKG, "lib", "variants", "libmistral_hip_*.so"))):
        n = os.path.basename(f)[len("libmistral_hip_"):-3]
        if not filt or any(s in n for s in filt):
            libs.append((n, f, {}))
            if "clk" in n and os.environ.get("PROBE_ZEROS"):
                libs.append((n + "+zeros", f, {"PROBE_ZERO": "1"}))
            if n.startswith("g_abl_mfmaonly"):  # the MFMA-only ablations also with both wave groups in lockstep
                libs.append((n + "+nostagger", f, {"MI_GEMM_STAGGER": "0"}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "prefill_probe.log"), "w")
    ref = {}
    for rep in range(reps):
        for name, path, extra in libs:
            env = dict(os.environ, **extra)
            if rep == 0 and name == "main":
                env["PROBE_CHECK"] = "1"
            if path:
                env["MISTRAL_HIP_LIB"] = path
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, timeout=240)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                if not line:
                    raise RuntimeError((r.stdout + r.stderr)[-300:])
                for l in r.stdout.splitlines():
                    if l.startswith("CHECK "):
                        print(l, flush=True)
                        log.write(l + "\n")
                d = json.loads(line[0][7:])
            except Exception as e:  # a variant that hangs or crashes must not cost the whole call
                msg = f"{name:22s} FAILED {str(e)[-200:]}"
                print(msg, flush=True)
                log.write(msg + "\n")
                continue
            if name == "main" and not ref:
                ref = {k: v["sha"] for k, v in d.items()}
            layer = sum(v["us"] for v in d.values())
            cols = "  ".join(f"{k} {v['us']:7.1f}us {v['tf']:6.0f}TF {'=' if v['sha'] == ref.get(k) else ('x' if v['finite'] else 'NaN')}" for k, v in d.items())
            msg = f"{name:30s} {cols}  | sum {layer:7.1f}"
            if any(v.get("clk") for v in d.values()):
                msg += "\n" + " " * 31 + "main loop of block 0, s_memtime ticks / 100 MHz ticks -> GHz: " + "  ".join(
                    f"{k} {v['clk'][0]}/{v['clk'][1]} = {v['clk'][0] / max(v['clk'][1], 1) / 10:.3f}" for k, v in d.items() if v.get("clk"))
            print(msg, flush=True)
            log.write(msg + "\n")
            log.flush()


if __name__ == "__main__":
    worker() if "--worker" in sys.argv else main()
