#!/usr/bin/env python3
"""Headline benchmark: batch-1 decode tokens/s (+ prefill tokens/s at 4k context) of the forward_partial hot path.

Workload (BASELINE.json configs[1]): Mistral-7B-v0.3 dimensions, all 32 layers, random-init bf16 weights,
sliding_window=4096, one 4096-token prefill, then greedy batch-1 decode.  A "step" is one decode token:
`Transformer.forward(next_token, [1], cache)` = one `mi_forward` call = 32 x 5 weight-streaming launches + LM head,
with the ring full (4096 keys per layer).  Inputs (weights, K/V rings, token ids) are resident in HBM.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # pipeline stages over RCCL

N > 1 (one rank per GPU, layer ranges as pipeline stages): the headline `value` stays the BASELINE metric - ONE sequence relayed
through the N stages, what the reference's pipeline does (transformer.py:195-237; "scaling": "strong": the same work on more
GPUs, and by construction no faster than on one).  The pipeline's THROUGHPUT mode - N independent sequences in flight, one per
stage at any time (mistral_inference/pipeline_decode.py) - is reported beside it under `pipeline_throughput`.

N = 1 additionally carries, as sub-objects of the same line: `nemo` and `mixtral` (BASELINE configs[2] and [3] measured by this
same script in a subprocess each), `parity` (the configs[0] model against the CPU oracle in this run) and `cpu_baseline`.

Prints ONE JSON line on rank 0 (see README/DESIGN.md section 6 for the roofline / cpu_baseline definitions).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

MISTRAL_7B = dict(dim=4096, n_layers=32, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                  vocab_size=32768, rope_theta=1e6, sliding_window=4096)
# The other BASELINE.json configs.  They are parity / evidence runs, not the headline line:
# `python bench.py --model nemo-12b --prefill 8192`, `--model mixtral-8x7b` (93 GB of random-init weights, fits one GPU).
PRESETS = {
    "mistral-7b": (MISTRAL_7B, "Mistral-7B-v0.3"),
    "nemo-12b": (dict(dim=5120, n_layers=40, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                      vocab_size=131072, rope_theta=1e6), "Mistral-Nemo-12B"),
    "mixtral-8x7b": (dict(dim=4096, n_layers=32, head_dim=128, hidden_dim=14336, n_heads=32, n_kv_heads=8, norm_eps=1e-5,
                          vocab_size=32000, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2)),
                     "Mixtral-8x7B"),
    # BASELINE configs[4]: 281 GB of bf16 weights - needs the 8 pipeline stages north_star names (35 GB per stage)
    "mixtral-8x22b": (dict(dim=6144, n_layers=56, head_dim=128, hidden_dim=16384, n_heads=48, n_kv_heads=8, norm_eps=1e-5,
                           vocab_size=32768, rope_theta=1e6, moe=dict(num_experts=8, num_experts_per_tok=2)),
                      "Mixtral-8x22B"),
}


def init_weights_(model, seed: int) -> None:
    """Random init of the reference tests' kind (U(+-1/sqrt(fan_in)) linears, N(0,1) embedding, unit norms),
    generated on the device tensor by tensor."""
    g = torch.Generator(device=model.device).manual_seed(seed)
    for name, p in model.named_parameters():
        with torch.no_grad():
            if name.endswith("norm.weight"):
                p.fill_(1.0)
            elif name.startswith("tok_embeddings"):
                p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
            else:
                bound = 1.0 / math.sqrt(p.shape[1])
                for r0 in range(0, p.shape[0], 8192):  # bounded fp32 temporaries
                    blk = p[r0:r0 + 8192]
                    blk.copy_((torch.rand(blk.shape, generator=g, device=p.device, dtype=torch.float32) * 2 - 1) * bound)


def build_model(params: dict, rank: int, world: int, device: str):
    from mistral_inference.args import TransformerArgs
    from mistral_inference.transformer import Transformer
    args = TransformerArgs.from_dict(params)
    args.max_batch_size = 1
    with torch.device("meta"):
        model = Transformer(args, pipeline_rank=rank, num_pipeline_ranks=world)
    model = model.to(torch.bfloat16).to_empty(device=device)
    init_weights_(model, seed=42 + rank)
    model._backend.invalidate()
    return model.eval()


def decode_bytes_per_token(p: dict, ctx: int, head: bool = True) -> int:
    """weights read once + K/V window read once (bf16).  head=False: a pipeline stage without the LM head."""
    D, L, H, Hkv, Dh, F, V = p["dim"], p["n_layers"], p["n_heads"], p["n_kv_heads"], p["head_dim"], p["hidden_dim"], p["vocab_size"]
    moe = p.get("moe")
    ffn = (moe["num_experts_per_tok"] * 3 * D * F + moe["num_experts"] * D) if moe else 3 * D * F
    per_layer = D * H * Dh + 2 * D * Hkv * Dh + H * Dh * D + 2 * D + ffn
    w = 2 * (L * per_layer + (V * D + D if head else 0))
    W = p.get("sliding_window") or ctx
    kv = L * 2 * min(ctx, W) * Hkv * Dh * 2
    return w + kv


def prefill_flops(p: dict, T: int) -> float:
    D, L, H, Hkv, Dh, F, V = p["dim"], p["n_layers"], p["n_heads"], p["n_kv_heads"], p["head_dim"], p["hidden_dim"], p["vocab_size"]
    moe = p.get("moe")
    lin = D * H * Dh + 2 * D * Hkv * Dh + H * Dh * D + (moe["num_experts_per_tok"] if moe else 1) * 3 * D * F
    W = p.get("sliding_window") or T
    pairs = sum(min(i + 1, W) for i in range(T))
    return 2.0 * T * (L * lin + V * D) + L * 4.0 * H * Dh * pairs


def _pmc_traffic(kernel_substr: str, dims_ok: bool):
    """HBM bytes per launch of the dominant kernel from the PMC counters: collected in separate `rocprofv3 --pmc
    FETCH_SIZE` / `--pmc WRITE_SIZE` passes (scripts/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes for
    gfx950 (2 x FETCH_SIZE), committed under profiles/.  Only valid for the named model dimensions and kernel."""
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    if dims_ok and os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        if kernel_substr in rec.get("kernel", ""):
            return rec["hbm_bytes_per_launch"], rec["source"]
    return None, None


def engine_roofline(model, cache, nxt, params: dict, iters: int, timed_us: float = None, timed_ctx: int = None, timed_steps: int = 0) -> dict:
    """The dominant kernel of the timed path is the persistent decode engine: ONE `decode_engine_kernel` launch per
    token streams every local layer's weights, the K/V rings and the LM head (and does the step's bookkeeping: position,
    embedding row, greedy sample - there is no other launch in a step).  Algorithmic bytes per launch = SURVEY.md 8(d)'s
    bytes per token at this context; launch duration = HIP events on the launch stream around the K TIMED steps themselves
    (`timed_us`, taken in timed_run: the graph-replayed greedy steps whose wall time is `ms_per_step`) - what
    `rocprofv3 --kernel-trace --stats` reports for the kernel in that loop (profiles/).  Two eager loops of `iters` steps are
    timed beside it to say where the difference to round 5's figure (an eager `forward()` loop) goes: `forward()` launches
    carry no sample (no logits stash, no 256 -> 1 gather of the workgroups' (max, argmax, sum-exp) behind the LM head - the
    kernel ends when workgroup 0 has finished that hop) and re-read one token id; the eager greedy session isolates that
    epilogue from the cost of replaying a one-kernel hipGraph per step."""
    dev = model.device
    stream = torch.cuda.current_stream(dev)
    for _ in range(2):
        nxt = torch.argmax(model.forward(nxt, [1], cache), dim=-1)
    ctx0 = cache._seen[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ids = nxt.clone()
    e0.record(stream)
    for _ in range(iters):
        model.forward(ids, [1], cache)   # same token id every step: no argmax kernel between the launches
    e1.record(stream)
    e1.synchronize()
    fwd_us = e0.elapsed_time(e1) * 1e3 / iters
    def session_us(**kw):
        sess = model.greedy_session(cache, ids, **kw)
        if kw.get("graph"):
            sess.engine_graph = True   # (keep the one-kernel hipGraph: what generate() did until round 5)
        sess.run(3)
        e0.record(stream)
        sess.run(iters)
        e1.record(stream)
        e1.synchronize()
        sess.collect()
        return e0.elapsed_time(e1) * 1e3 / iters
    greedy_eager_us = session_us(graph=False)  # the same steps as the timed loop, launched plainly
    greedy_graph_us = session_us(graph=True)   # ... and replayed from a one-kernel hipGraph per step
    us, ctx_mid, n_timed = (timed_us, timed_ctx, timed_steps) if timed_us else (fwd_us, ctx0 + iters // 2, iters)
    bytes_per_launch = decode_bytes_per_token(params, ctx_mid)
    gbs = bytes_per_launch / (us * 1e-6) / 1e9
    dims_ok = ((model.args.dim, model.args.hidden_dim, model.args.n_layers) == (MISTRAL_7B["dim"], MISTRAL_7B["hidden_dim"], MISTRAL_7B["n_layers"])
               and not params.get("moe"))
    traffic, traffic_src = _pmc_traffic("decode_engine_kernel", dims_ok)
    group = model.args.n_heads // model.args.n_kv_heads
    return {"bound": "hbm", "kernel": f"decode_engine_kernel<{group}> (persistent: all layers + LM head + sample of one decode step in one launch)",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_static": True, "traffic_source": traffic_src, "bytes_per_launch": bytes_per_launch,
            "avg_launch_us": round(us, 2), "launches_timed": n_timed,
            "timing": ("HIP events on the launch stream around the TIMED greedy steps themselves (one kernel per step)" if timed_us
                       else "HIP events around an eager forward() loop"),
            "other_loops_us": {"eager_forward_no_sample": round(fwd_us, 2), "greedy_session_plain_launches": round(greedy_eager_us, 2),
                               "greedy_session_one_kernel_hipgraph": round(greedy_graph_us, 2), "iters": iters,
                               "reading": "greedy - forward = the fused sample's hop behind the LM head; hipgraph - plain = what "
                                          "replaying a one-kernel hipGraph per step costs over queued plain launches (why engine "
                                          "sessions launch plainly since round 6)"}}


def dominant_kernel_roofline(model, iters: int) -> dict:
    """Launch path: the W1|W3 gate/up GEMV (54 % of the decode bytes): algorithmic bytes per launch / average launch
    duration, timed live with HIP events on the launch stream, cycling through all local layers' weights so
    no launch re-reads what the previous one left in the 256 MiB Infinity Cache."""
    from mistral_inference import _hip
    a = model.args
    dev = model.device
    x = torch.randn(1, a.dim, device=dev).to(torch.bfloat16)
    out = torch.empty(1, a.hidden_dim, device=dev, dtype=torch.bfloat16)
    blocks = list(model.layers.values())
    stream = torch.cuda.current_stream(dev)

    def run(n):
        for i in range(n):
            b = blocks[i % len(blocks)]
            _hip.linear(x, (b.feed_forward.w1.weight, b.feed_forward.w3.weight), _hip.EPI_SWIGLU,
                        norm_w=b.ffn_norm.weight, eps=a.norm_eps, out=out)

    run(len(blocks))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = iters * len(blocks)
    e0.record(stream)   # torch's current stream IS the stream mi_linear launches on (see _hip.stream_ptr)
    run(n)
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    bytes_per_launch = 2 * a.hidden_dim * a.dim * 2 + 2 * a.dim * 2 + a.hidden_dim * 2
    gbs = bytes_per_launch / (us * 1e-6) / 1e9
    traffic, traffic_src = _pmc_traffic("gemv_kernel", (a.dim, a.hidden_dim) == (MISTRAL_7B["dim"], MISTRAL_7B["hidden_dim"]))
    return {"bound": "hbm", "kernel": "gemv_kernel<1,SWIGLU> (RMSNorm + W1|W3 GEMV + SiLU*mul)", "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_static": True, "traffic_source": traffic_src, "bytes_per_launch": bytes_per_launch, "avg_launch_us": round(us, 2),
            "launches_timed": n}


def stage_roofline(model, cache, params: dict, rank: int, world: int, iters: int) -> dict:
    """N > 1: this rank's stage alone - `iters` back-to-back decode calls of its layer range without the hops (what
    pipeline_decode.InterleavedDecoder issues per tick), HIP events on the launch stream, against the stage's own bytes."""
    dev = model.device
    be = model._backend
    last = rank == world - 1
    h = torch.zeros((1, model.args.dim), dtype=model.dtype, device=dev)
    ids = torch.ones(1, dtype=torch.long, device=dev) if rank == 0 else None
    logits = torch.empty((1, model.vocab_size), dtype=torch.float32, device=dev) if last else None
    stream = torch.cuda.current_stream(dev)

    def one():
        meta = cache.batch_metadata([1])
        be.run_stack(model, h, ids, meta, cache, logits)
        cache.advance_host([1])

    for _ in range(3):
        one()
    ctx0 = cache._seen[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        one()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    nbytes = decode_bytes_per_token(dict(params, n_layers=model.n_local_layers), ctx0 + iters // 2, head=last)
    gbs = nbytes / (us * 1e-6) / 1e9
    return {"rank": rank, "layers": model.n_local_layers, "lm_head": last, "bytes_per_call": nbytes, "avg_call_us": round(us, 2),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}


def reference_baseline(params: dict, ctx: int, steps: int = 6):
    """The UNMODIFIED reference timed on this box's host cores (oracle/time_reference.py in a subprocess: the reference
    shares the product's package name).  None where the reference source is absent (the GPU box)."""
    import subprocess
    # the source tree where it exists (the build container), else its byte-compiled form oracle/build_ref.py put under
    # oracle/_ref/ (git-ignored, travels with the working tree to the GPU box like the built .so does)
    ref = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")
    if not os.path.isdir(os.path.join(ref, "mistral_inference")):
        ref = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref, "mistral_inference")):
        return None
    p = {k: v for k, v in params.items()}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_reference.py"), "--params", json.dumps(p),
                            "--ctx", str(ctx), "--steps", str(steps)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MISTRAL_REFERENCE_SRC=ref))
        if r.returncode != 0:
            return None
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["reference_form"] = "source tree" if ref.endswith("/src") else "oracle/_ref (byte-compiled from the unmodified source by oracle/build_ref.py)"
        return out
    except Exception:  # noqa: BLE001  (a baseline that cannot be taken is reported as absent, never as a number)
        return None


def cpu_baseline(params: dict, ctx: int, steps: int = 6) -> dict:
    """CPU baseline on this box's host cores, bounded sample (the same decode step at the same context with 2 of the
    layers + LM head, scaled linearly in the layer count): the unmodified reference where its source is present
    (`kind: "reference"`, with the oracle port's number alongside), else the oracle port (`kind: "port"`)."""
    ref = reference_baseline(params, ctx, steps)
    port = port_baseline(params, ctx, steps)
    if ref is not None:
        ref["port_value"] = port["value"]
        return ref
    # The GPU box has no reference source.  The unmodified reference was timed once, on the build container, with the same
    # bounded sample (Mistral-7B dims): carried along so that the record holds both numbers.
    stored = os.path.join(ROOT, "profiles", "r02_cpu_baseline_reference_vs_port.json")
    if params.get("dim") == 4096 and params.get("n_layers") == 32 and not params.get("moe") and os.path.exists(stored):
        try:
            r = json.load(open(stored))["reference"]
            port["reference_container_value"] = r["value"]
            port["reference_container_cores"] = r["cores"]
            port["reference_container_source"] = "profiles/r02_cpu_baseline_reference_vs_port.json (build container, no GPU)"
        except (KeyError, ValueError, OSError):
            pass
    return port


def port_baseline(params: dict, ctx: int, steps: int = 6) -> dict:
    """The oracle (CPU restatement of the reference, oracle/mistral_oracle.py) timed the same way."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.nn.functional as F
    import mistral_oracle as mo
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nl = 1 if params.get("moe") else 2   # (an 8-expert layer of the 8x7B dims is 2.9 GB of host weights)
    p2 = dict(params, n_layers=nl)
    oargs = mo.OracleArgs.from_params(p2)
    w = mo.synth_weights(oargs, seed=1)
    om = mo.OracleModel(oargs, w)
    W = params.get("sliding_window") or ctx
    MAX_STEPS = 200
    cache = mo.OracleCache(nl, 1, ctx + MAX_STEPS + 4, oargs.n_kv_heads, oargs.head_dim, params.get("sliding_window"),
                           dtype=torch.bfloat16)
    for l in range(nl):  # a full ring, as after the 4096-token prefill
        cache.k[l].copy_(torch.randn(cache.k[l].shape).to(torch.bfloat16))
        cache.v[l].copy_(torch.randn(cache.v[l].shape).to(torch.bfloat16))
    cache.seen = [ctx]
    tok = torch.tensor([1])
    # torch's intra-op pool does not scale a batch-1 decode to hundreds of host threads: try a few pool sizes on
    # one step each and keep the fastest (that count is what "cores" reports)
    best = (float("inf"), 1)
    with torch.inference_mode():
        for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            om.forward(tok, [1], cache)
            t0 = time.perf_counter()
            om.forward(tok, [1], cache)
            best = min(best, (time.perf_counter() - t0, n))
    cores = best[1]
    torch.set_num_threads(cores)
    # the bounded sample: about 6 s of decode steps + the LM-head loop (~10 s of CPU work in all), never fewer than `steps`
    steps = max(steps, min(MAX_STEPS, int(6.0 / max(best[0], 1e-3))))
    cache.seen = [ctx]
    with torch.inference_mode():
        om.forward(tok, [1], cache)
        t0 = time.perf_counter()
        for _ in range(steps):
            om.forward(tok, [1], cache)
        t_full = (time.perf_counter() - t0) / steps
        h = torch.randn(1, oargs.dim).to(torch.bfloat16)
        t0 = time.perf_counter()
        for _ in range(steps):
            F.linear(mo.rms_norm(h, w["norm.weight"], 1e-5), w["output.weight"]).float()
        t_head = (time.perf_counter() - t0) / steps
    per_layer = max(1e-9, (t_full - t_head) / nl)
    t_model = t_head + params["n_layers"] * per_layer
    return {"value": round(1.0 / t_model, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle decode step at ctx {ctx} (W={W}) with {nl} of {params['n_layers']} layers + LM head, "
                      f"{steps} steps, bf16, {cores} threads; per-layer time x{params['n_layers']} + head "
                      f"({per_layer * 1e3:.1f} ms/layer, {t_head * 1e3:.1f} ms head)"}


def sub_measurement(model_key: str, prefill: int, steps: int, warmup: int, timeout_s: int = 420, extra=()) -> dict:
    """Another BASELINE config measured by THIS script in a subprocess of its own (its own weights, its own timing bracket;
    a crash, a time-out or an unexpected line there costs the sub-object, never the headline line).  Returns the fields a
    reader needs.  `extra`: further command-line flags (`--layers 7`: one pipeline stage; `--batch 3`: the mistral-demo shape)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--model", model_key, "--prefill", str(prefill), "--steps", str(steps),
           "--warmup", str(warmup), "--no-cpu-baseline", "--no-extras", *extra]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rf = d.get("roofline", {})
        out = {"workload": d["config"]["workload"], "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
               "warmup": d["warmup"], "context_at_timing": d["config"]["context_at_timing"],
               "hbm_roofline_frac": d["hbm_roofline_step"]["frac"], "decode_launch": d["config"]["decode_launch"]}
        if "bytes_per_token" in d["hbm_roofline_step"]:
            out["bytes_per_token"] = d["hbm_roofline_step"]["bytes_per_token"]
        else:  # --batch B: weights once + B ring windows per step
            out["bytes_per_step"] = d["hbm_roofline_step"]["bytes_per_step"]
            out["per_sequence_tokens_per_s"] = d.get("per_sequence_tokens_per_s")
        if rf:
            out["dominant_kernel"] = {"kernel": rf.get("kernel"), "frac": rf.get("frac"), "avg_launch_us": rf.get("avg_launch_us")}
        if "prefill" in d:
            out.update(prefill_tokens=d["prefill"]["tokens"], prefill_tokens_per_s=d["prefill"]["tokens_per_s"],
                       prefill_mfma_frac=d["prefill"]["mfma_frac"])
        out["wall_s"] = round(time.perf_counter() - t0, 1)
        return out
    except Exception as e:  # noqa: BLE001  (reported, never hidden: a missing key is a failure of the sub-object only)
        return {"error": f"{type(e).__name__}: {str(e)[:300]}", "wall_s": round(time.perf_counter() - t0, 1)}


def parity_in_this_run() -> dict:
    """SURVEY.md 8(d) "parity in the same run": BASELINE configs[0] (Mistral-7B dims, 2 layers, 32-token prompt + 16 greedy
    tokens) on the HIP path against the CPU oracle on identical weights and prompt.  The comparison itself lives in
    __graft_entry__.parity_numbers (next to smoke(), which holds the oracle as its checker)."""
    try:
        import __graft_entry__ as ge
        return ge.parity_numbers()
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {str(e)[:300]}"}


def timed_run(opt, params: dict, rank: int, world: int, dev: str, T0: int, K: int, Wm: int, sync):
    """Build the model (this rank's pipeline stage), run one untimed and one timed T0-token prefill, W warm-up decode steps,
    then time exactly K decode steps between two synchronisations (+ barriers).  Returns the model, its cache, the last
    token, the decode seconds and the prefill seconds (both the maximum over ranks)."""
    from mistral_inference.cache import BufferCache
    model = build_model(params, rank, world, dev)
    a = model.args
    cache = BufferCache(model.n_local_layers, 1, T0 + K + max(Wm, 2) + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device=dev,
                        dtype=torch.bfloat16)
    cache.reset()
    prompt = torch.randint(0, a.vocab_size, (T0,), generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.inference_mode():
        # ---- prefill: one untimed pass (allocates the workspace / logits buffers), then the timed pass on a
        # reset cache; includes the [T, V] fp32 LM head the API contract requires
        logits = model.forward(prompt, [T0], cache)
        del logits
        prefill_all = []
        for _ in range(3):  # three timed passes, the MEDIAN is reported (a single sample of a power-capped quantity is noise)
            cache.reset()
            sync()
            t0 = time.perf_counter()
            logits = model.forward(prompt, [T0], cache)
            sync()
            prefill_all.append(time.perf_counter() - t0)
            nxt = torch.argmax(logits[-1:], dim=-1)
            del logits
        prefill_s = sorted(prefill_all)[1]
        # ---- decode: the generate() loop body - forward(next_token, [1], cache) under the decode hipGraph context
        # (capture happens inside the warm-up steps; W >= 2 keeps it out of the timed region)
        if opt.loop == "greedy":
            # generate()'s temperature-0 loop body (mistral_inference/generate.py -> Transformer.greedy_session): one
            # native call per token - argmax + log-softmax are the LM head's epilogue, the sample feeds the next step on
            # the device - replayed from a hipGraph.  Nothing of the step is skipped: logits [1, V] are written every step.
            sess = model.greedy_session(cache, nxt, graph=not opt.no_graph)
            W = max(Wm, 2)
            # W untimed steps; the health check (collect = a synchronisation, a device->host copy and some Python: the GPU idles
            # for milliseconds) sits behind the FIRST one, so that the rest of the warm-up runs back to back into the bracket's
            # synchronisation.  scripts/first_step_probe.py: behind >= ~2 ms of idle time the first FOUR steps run slow (2793,
            # 2958, 2788, 2695 us against 2615: the clocks ramp over ~10 ms), behind <= 1 ms only the first, by 25 us.  Rounds 5-6
            # had the check before the last two steps: ~0.3 ms of that ramp was charged to the K timed steps (15 us per step at K = 20).
            if W > 2:
                sess.run(1)
                sess.collect()
                sess.run(W - 1)
            else:
                sess.run(W)
            dt, left, collect_s, ev_ms = 0.0, K, 0.0, 0.0
            stream = torch.cuda.current_stream(dev)  # (the stream mi_forward launches on and the captured step is replayed on)
            # (events are created BEFORE the bracket's synchronisation: whatever the host does between that synchronisation and the
            # first launch is idle GPU time inside the timed region)
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K // sess.HIST + 2)]
            sync()
            while left > 0:                      # (the session's history ring holds 1024 steps between collects)
                n = min(left, sess.HIST - sess._pending)  # (the two warm-up steps above are still uncollected)
                e0, e1 = events.pop()
                t0 = time.perf_counter()
                e0.record(stream)                # HIP events on the launch stream around the SAME steps the wall clock brackets:
                sess.run(n)                      # on the persistent engine a step is one kernel, so this is the dominant kernel's
                e1.record(stream)                # launch-to-launch time inside the timed loop (roofline.avg_launch_us)
                sync()
                t1 = time.perf_counter()
                dt += t1 - t0
                ev_ms += e0.elapsed_time(e1)
                toks, _ = sess.collect()         # verifies that the device completed every step (and which path ran)
                collect_s += time.perf_counter() - t1   # generate() pays this once per chunk (32 steps with an eos_id, else 1024)
                left -= n
            nxt = toks[-1]
            timed_run.used_graph = bool(sess._use_graph)
            timed_run.collect_s = collect_s
            timed_run.event_us_per_step = ev_ms * 1e3 / K
        else:
            # the sampling loop's body (temperature > 0, or pipeline stages): forward() under the decode hipGraph + torch.argmax
            ctx = contextlib.nullcontext() if opt.no_graph else model.graphed_decode(cache)
            with ctx:
                for _ in range(max(Wm, 2)):
                    nxt = torch.argmax(model.forward(nxt, [1], cache), dim=-1)
                sync()
                t0 = time.perf_counter()
                for _ in range(K):
                    nxt = torch.argmax(model.forward(nxt, [1], cache), dim=-1)
                sync()
                dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt, prefill_s], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt, prefill_s = tmax.tolist()
    timed_run.prefill_all = prefill_all
    return model, cache, nxt, dt, prefill_s


def batch_run(opt, params: dict, B: int, dev: str, T0: int, K: int, Wm: int) -> dict:
    """`--batch B` (evidence run, not the headline): B sequences decoded together, as `mistral-demo` does with its three prompts
    (reference main.py:124,220) - the generate() loop's fused session at batch B on the launch path (the persistent engine is a
    batch-1 kernel).  Every weight row is streamed once per step and reduced against B activation rows; each sequence reads its
    own K/V ring: bytes per step = weights + B x ring window.  One JSON object; `value` = B tokens per step / step time."""
    from mistral_inference.args import TransformerArgs
    from mistral_inference.cache import BufferCache
    from mistral_inference.transformer import Transformer
    from mistral_inference import _hip
    args = TransformerArgs.from_dict(params)
    args.max_batch_size = B
    with torch.device("meta"):
        model = Transformer(args)
    model = model.to(torch.bfloat16).to_empty(device=dev)
    init_weights_(model, seed=42)
    model._backend.invalidate()
    model.eval()
    a = model.args
    cache = BufferCache(model.n_local_layers, B, T0 + K + max(Wm, 2) + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device=dev,
                        dtype=torch.bfloat16)
    cache.reset()
    g = torch.Generator().manual_seed(0)
    prompts = torch.randint(0, a.vocab_size, (B * T0,), generator=g).to(dev)
    with torch.inference_mode():
        logits = model.forward(prompts, [T0] * B, cache)  # one batch of B x T0 prompt tokens
        ends = torch.arange(1, B + 1, device=dev) * T0 - 1
        nxt = torch.argmax(logits.index_select(0, ends), dim=-1)
        del logits
        sess = model.greedy_session(cache, nxt, graph=not opt.no_graph)
        W = max(Wm, 2)
        sess.run(1)        # (the health check behind the first warm-up step, the rest back to back into the bracket: timed_run)
        sess.collect()
        sess.run(W - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sess.collect()
    st = _hip.decode_engine_status(model._backend._workspace)
    ctx_len = T0 + W + K // 2
    w_bytes = decode_bytes_per_token(params, ctx_len)
    Wn = params.get("sliding_window") or ctx_len
    kv_one = params["n_layers"] * 2 * min(ctx_len, Wn) * params["n_kv_heads"] * params["head_dim"] * 2
    step_bytes = w_bytes + (B - 1) * kv_one
    gbs = step_bytes / (dt / K) / 1e9
    return {"metric": f"decode tokens/sec/GPU (batch={B}, seq=1)", "value": round(B * K / dt, 2), "unit": "tokens/s", "n_gpus": 1,
            "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{PRESETS[opt.model][1]} dims, {params['n_layers']} layers, random-init bf16, {B} x {T0}-token prefill then "
                                   f"batch-{B} greedy decode, sliding_window={params.get('sliding_window')}",
                       "batch": B, "prefill_tokens": T0, "context_at_timing": ctx_len,
                       "decode_launch": ("persistent decode engine" if st["engine_launches"] > 0 else "6 launches per layer (launch path)")
                       + ", greedy sample fused into the step, " + ("hipGraph replay" if sess._use_graph else "plain launches")},
            "hbm_roofline_step": {"bytes_per_step": step_bytes, "weights_once_plus_kv_per_sequence": [w_bytes - kv_one, kv_one],
                                  "achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "per_sequence_tokens_per_s": round(K / dt, 2)}


def interleaved_run(opt, model, rank: int, world: int, dev: str, T0: int, K: int, Wm: int, sync):
    """N > 1: the pipeline's THROUGHPUT - `world` independent sequences, one per stage at any time
    (mistral_inference/pipeline_decode.py): every stage runs one batch-1 decode call per tick on a different sequence, the
    results move one stage along the ring.  Every sequence gets its own K/V rings and the same T0-token prefill as the
    single-stream run; W untimed rounds, then exactly K timed rounds (a round = one new token for every sequence) between
    two barriers + synchronisations.  Returns (seconds, maximum over ranks)."""
    from mistral_inference.cache import BufferCache
    from mistral_inference.pipeline_decode import InterleavedDecoder
    a = model.args
    prompt = torch.randint(0, a.vocab_size, (T0,), generator=torch.Generator().manual_seed(0)).to(dev)
    caches, first = [], []
    with torch.inference_mode():
        for j in range(world):
            c = BufferCache(model.n_local_layers, 1, T0 + K + max(Wm, 2) + 64, a.n_kv_heads, a.head_dim, a.sliding_window, device=dev,
                            dtype=torch.bfloat16)
            c.reset()
            logits = model.forward(prompt, [T0], c)
            first.append(torch.argmax(logits[-1:], dim=-1))
            del logits
            caches.append(c)
        dec = InterleavedDecoder(model, caches, torch.cat(first))
        dec.run(max(Wm, 2))
        sync()
        t0 = time.perf_counter()
        dec.run(K)
        sync()
        dt = time.perf_counter() - t0
    interleaved_run_graph[0] = bool(dec._use_graph and dec.ticks_replayed > 0)
    tmax = torch.tensor([dt, dec.tick_host_us], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    del dec, caches
    interleaved_run.tick_host_us = float(tmax[1].item())
    interleaved_run.tick_form = "hipGraph replay (stage call + grouped exchange per tick)" if interleaved_run_graph[0] else "eager"
    return float(tmax[0].item())


interleaved_run_graph = [False]


def respawn_under_torchrun(n: int) -> int:
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface (the container hostname may not resolve)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prefill", type=int, default=4096, help="prompt tokens (BASELINE configs[1]: 4096)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers => NOT the named config")
    ap.add_argument("--model", default="mistral-7b", choices=sorted(PRESETS), help="default = BASELINE configs[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue decode steps launch by launch (no hipGraph replay)")
    ap.add_argument("--loop", default="greedy", choices=["greedy", "forward"],
                    help="greedy: generate()'s temperature-0 loop (sample fused into the step); forward: forward() + torch.argmax per token")
    ap.add_argument("--no-mixtral", action="store_true", help="N > 1: skip the Mixtral sub-measurement")
    ap.add_argument("--batch", type=int, default=1, help="evidence run: B sequences decoded together (mistral-demo runs 3); 1 = the headline")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the `nemo` / `mixtral` / `parity` sub-objects (the headline fields are the same either way)")
    ap.add_argument("--mixtral-layers", type=int, default=None, help="debug only: layers of the N > 1 Mixtral sub-measurement")
    opt = ap.parse_args()

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, RCCL over xGMI) exactly as the
        # documented torch.distributed.run line would, and let rank 0 of that job print the JSON line
        sys.exit(respawn_under_torchrun(opt.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == opt.gpus, f"--gpus {opt.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    local = local % torch.cuda.device_count()  # (test rigs may run several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        # "nccl" is RCCL on ROCm (xGMI between the GPUs of a node).  MI_DIST_BACKEND=gloo exists only so that the
        # pipeline path can be exercised on a single-GPU box (RCCL refuses two ranks on one device).
        torch.distributed.init_process_group(os.environ.get("MI_DIST_BACKEND", "nccl"))

    params, model_name = PRESETS[opt.model]
    params = dict(params)
    if opt.layers:
        params["n_layers"] = opt.layers
    T0, K, Wm = opt.prefill, opt.steps, opt.warmup

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if opt.batch > 1:
        assert world == 1, "--batch is a single-GPU evidence run"
        print(json.dumps(batch_run(opt, params, opt.batch, dev, T0, K, Wm)), flush=True)
        return

    model, cache, nxt, dt, prefill_s = timed_run(opt, params, rank, world, dev, T0, K, Wm, sync)
    # N > 1: a single sequence is a relay through the stages (dt above: N GPUs decode no faster than one); the pipeline's
    # throughput is measured with one sequence per stage in flight.  MI_BENCH_INTERLEAVE=0 skips it.
    interleave = world > 1 and opt.loop == "greedy" and os.environ.get("MI_BENCH_INTERLEAVE", "1") != "0"
    dt_il = None
    if interleave:
        ok = 1
        try:
            dt_il = interleaved_run(opt, model, rank, world, dev, T0, K, Wm, sync)
        except Exception as e:  # keep the single-stream line instead of no line
            print(f"[bench] rank {rank}: interleaved measurement failed ({type(e).__name__}: {e}); reporting the single-stream relay only",
                  file=sys.stderr, flush=True)
            ok = 0
        # every rank learns whether ALL ranks finished.  This covers failures that every rank sees (a shape the throughput mode
        # declines, an allocation that fails everywhere); a rank that raises ALONE inside interleaved_run leaves its peers in
        # that function's collectives, and the job ends with the process group's time-out, not with a fallback line
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag.item()) == 0:
            interleave, dt_il = False, None

    from mistral_inference import _hip
    engine = _hip.decode_engine_status(model._backend._workspace)
    assert engine["status"] == 0 and engine["bad_id"] == 0, f"device-side error flags: {engine}"

    def decode_launch_label() -> str:
        # the engine counts its own launches in the workspace: that is how we know which path was timed
        kind = "persistent decode engine (1 launch per token)" if engine["engine_launches"] > 0 else "6 launches per layer"
        kind += ", greedy sample fused into the step" if opt.loop == "greedy" else ", forward() + torch.argmax"
        if world > 1 and opt.loop == "greedy":
            kind += " (one session per stage; the sample returns to stage 0 as 8 bytes per token, no logits broadcast)"
        from mistral_inference.distributed import RcclComm
        eager = opt.no_graph or (world > 1 and not isinstance(model.pp_comm, RcclComm))
        if opt.loop == "greedy":  # what the session really did: an engine step is one kernel and is launched plainly
            eager = eager or not getattr(timed_run, "used_graph", True)
        return kind + (", plain launches queued back to back (no hipGraph)" if eager else ", hipGraph replay")

    def report() -> dict:
        ctx_len = T0 + Wm + K // 2
        step_bytes = decode_bytes_per_token(params, ctx_len)
        ms = dt / K * 1e3
        step_gbs = step_bytes / (dt / K) / 1e9
        out = {
            "metric": "decode tokens/sec/GPU (batch=1, seq=1)", "value": round(K / dt, 2), "unit": "tokens/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{model_name} dims, {params['n_layers']} layers, random-init bf16, "
                                   f"{T0}-token prefill then batch-1 greedy decode, sliding_window={params.get('sliding_window')}",
                       "batch": 1, "prefill_tokens": T0, "context_at_timing": ctx_len,
                       "decode_launch": decode_launch_label(),
                       "parallelism": "single GPU" if world == 1 else
                       f"pp{world} (layer ranges; {type(model.pp_comm).__name__} send/recv + logits broadcast, process group "
                       f"{torch.distributed.get_backend()})"},
            "hbm_roofline_step": {"bytes_per_token": step_bytes, "achieved_GBs": round(step_gbs, 1), "peak_GBs": HBM_PEAK_GBS,
                                  "frac": round(step_gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_6290": round(step_gbs / 6290.0, 4)},
            "prefill": {"tokens": T0, "seconds": round(prefill_s, 4), "timing": "median of 3 timed passes",
                        "seconds_all": [round(x, 4) for x in getattr(timed_run, "prefill_all", [])],
                        "tokens_per_s": round(T0 / prefill_s, 1),
                        "tflops": round(prefill_flops(params, T0) / prefill_s / 1e12, 1), "mfma_peak_tflops": 2500.0,
                        "mfma_frac": round(prefill_flops(params, T0) / prefill_s / 2.5e15, 4)},
        }
        if opt.loop == "greedy":
            # not inside the K-step bracket: the once-per-chunk read-back of the samples (status copy + host sync + gather)
            out["collect_ms_per_chunk"] = round(getattr(timed_run, "collect_s", 0.0) * 1e3, 3)
        if world > 1:
            # the record shows what the collective layer saw: process-group backend and size, and - with the C-ABI transport -
            # the communicator's own rank count
            out["distributed"] = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                                  "transport": type(model.pp_comm).__name__,
                                  "rccl_comm_ranks": getattr(model.pp_comm, "world_size", None) if type(model.pp_comm).__name__ == "RcclComm" else None}
            out["hbm_roofline_step"]["note"] = (f"ONE sequence relayed through {world} stages: the stages run one after the other, so the "
                                                f"roofline time is the one-GPU sum (SURVEY.md 8e) and frac is priced against ONE GPU's peak")
        if dt_il is not None:
            # N > 1: the headline above is the BASELINE metric (one sequence through the stages).  The throughput mode - `world`
            # sequences in flight, one per stage, each call still batch 1 / seq 1 on that sequence's own rings; a round = one
            # new token for EVERY sequence - is an extra of this implementation and lives in its own object.
            rate = K * world / dt_il
            agg = step_bytes * rate / 1e9
            out["pipeline_throughput"] = {
                "tokens_per_s": round(rate, 2), "tokens_per_s_per_gpu": round(rate / world, 2), "ms_per_round": round(dt_il / K * 1e3, 4),
                "sequences_in_flight": world, "scaling": "weak",
                "hbm_roofline_frac": round(agg / (HBM_PEAK_GBS * world), 4), "hbm_peak_GBs": HBM_PEAK_GBS * world,
                "tick_host_us": round(getattr(interleaved_run, "tick_host_us", float("nan")), 2),
                "tick_form": getattr(interleaved_run, "tick_form", "eager"),
                "note": "one sequence per stage in flight, ring of grouped send+recv per tick (mistral_inference/pipeline_decode.py); "
                        "tick_host_us = host time per tick of the loop (max over ranks)"}
        # the dominant kernel is timed on this rank's own layers (any N)
        if engine["engine_launches"] > 0 and world == 1:
            with torch.inference_mode():
                ev = getattr(timed_run, "event_us_per_step", None) if opt.loop == "greedy" else None
                out["roofline"] = engine_roofline(model, cache, nxt, params, iters=24, timed_us=ev, timed_ctx=ctx_len, timed_steps=K)
            if not params.get("moe"):
                out["launch_path_gemv_w13"] = dominant_kernel_roofline(model, iters=2)
        elif not params.get("moe"):
            out["roofline"] = dominant_kernel_roofline(model, iters=4)
        else:  # MoE on the launch path: no single dominant kernel is timed - the whole-step figure is the roofline number
            out["roofline"] = {"bound": "hbm", "kernel": "decode step (launch path: router + expert GEMVs)", "achieved": round(step_gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(step_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                               "bytes_per_launch": step_bytes}
        if not opt.no_cpu_baseline:
            # rank 0 only, at every N (the other ranks wait at the barrier below): the reference's own CPU path on this box's
            # host cores for the headline model - the whole model, whatever the number of stages the GPUs split it into
            out["cpu_baseline"] = cpu_baseline(params, T0)
        return out

    out = report() if rank == 0 else None
    if world > 1:  # the other ranks wait here while rank 0 times its dominant kernel
        torch.distributed.barrier()
        # every stage's own decode call (no hop: Backend.run_stack directly, as the throughput mode issues it) against the
        # bytes of its own layers (+ the LM head on the last stage)
        with torch.inference_mode():
            mine = stage_roofline(model, cache, params, rank, world, iters=16)
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
        if out is not None:
            out["roofline_per_rank"] = per_rank
    if world == 1 and opt.model == "mistral-7b" and not opt.layers and not opt.no_extras:
        # the other BASELINE configs and parity, in the one line the driver records (the headline fields above are final -
        # they go to stderr right away, so that a time-out or a crash in an extra cannot lose a measured headline)
        print("[bench] headline measured (sub-objects follow): " + json.dumps(out), file=sys.stderr, flush=True)
        del model, cache, nxt
        torch.cuda.empty_cache()
        out["parity"] = parity_in_this_run()
        if isinstance(out["parity"], dict):
            out["parity"]["scope"] = ("BASELINE configs[0] only (2 layers, 48 tokens), in this process; depth parity of configs[1]-[4] "
                                      "(32 layers vs the bf16 AND fp32 oracle, Nemo x 4 layers, Mixtral-8x7B x 4, 8x22B x 3) is "
                                      "`pytest -m gpu` (tests/test_gpu_depth.py), not this object")
        torch.cuda.empty_cache()
        # (the sub-objects time their own brackets of >= 64 steps behind >= 8 warm-up steps - this script's defaults - whatever the
        #  headline's K / W are, and say so in their `steps` / `warmup` fields: a 20-step bracket carries ~0.2 ms of bracket edges)
        Ks, Ws = max(K, 64), max(Wm, 8)
        out["nemo"] = sub_measurement("nemo-12b", 8192, Ks, Ws)
        out["mixtral"] = sub_measurement("mixtral-8x7b", T0, Ks, Ws)
        # BASELINE configs[4]: what EACH of the 8 pipeline stages of Mixtral-8x22B runs (7 of 56 layers, 35 GB; the last stage
        # adds the LM head, which this stage measurement carries) - the per-GPU number of the 8-stage deployment
        out["x22b_stage"] = sub_measurement("mixtral-8x22b", T0, Ks, Ws, extra=("--layers", "7"))
        # the mistral-demo shape: three prompts decoded together (reference main.py:124,220), launch path
        out["batch3"] = sub_measurement("mistral-7b", T0, Ks, Ws, extra=("--batch", "3"))
    if world > 1 and not opt.no_mixtral and opt.model == "mistral-7b" and (not opt.layers or opt.mixtral_layers):
        # north_star: "Mixtral-8x7B pipeline-parallel tokens/sec reported at 1/2/4/8 GPUs" (BASELINE configs[3], and
        # configs[4] - Mixtral-8x22B over 8 stages - where 8 GPUs are present).  The N = 1 headline line is untouched; a
        # multi-GPU run additionally measures the Mixtral model over the same N stages and reports it as a sub-object.
        del model, cache, nxt
        torch.cuda.empty_cache()
        mx_name = "mixtral-8x22b" if world >= 8 else "mixtral-8x7b"
        mx_params = dict(PRESETS[mx_name][0])
        if opt.mixtral_layers:
            mx_params["n_layers"] = opt.mixtral_layers
        m2, c2, _, dt2, pre2 = timed_run(opt, mx_params, rank, world, dev, T0, K, Wm, sync)
        dt2_il = interleaved_run(opt, m2, rank, world, dev, T0, K, Wm, sync) if interleave else None
        if rank == 0:
            ctx_len = T0 + Wm + K // 2
            b2 = decode_bytes_per_token(mx_params, ctx_len)
            out["mixtral"] = {"model": f"{PRESETS[mx_name][1]} dims, {mx_params['n_layers']} layers, random-init bf16, pp{world}",
                              "tokens_per_s": round(K / dt2, 2), "ms_per_step": round(dt2 / K * 1e3, 4), "bytes_per_token": b2,
                              "hbm_roofline_frac": round(b2 / (dt2 / K) / 1e9 / HBM_PEAK_GBS, 4),
                              "prefill_tokens_per_s": round(T0 / pre2, 1),
                              "prefill_mfma_frac": round(prefill_flops(mx_params, T0) / pre2 / 2.5e15 / world, 4),
                              "transport": type(m2.pp_comm).__name__}
            if dt2_il is not None:  # the pipeline's throughput (one sequence per stage in flight) beside the single-sequence relay
                r2 = K * world / dt2_il
                out["mixtral"]["pipeline_throughput"] = {
                    "tokens_per_s": round(r2, 2), "tokens_per_s_per_gpu": round(r2 / world, 2), "ms_per_round": round(dt2_il / K * 1e3, 4),
                    "sequences_in_flight": world, "hbm_roofline_frac": round(b2 * r2 / 1e9 / (HBM_PEAK_GBS * world), 4),
                    "tick_host_us": round(getattr(interleaved_run, "tick_host_us", float("nan")), 2)}
        del m2, c2
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
