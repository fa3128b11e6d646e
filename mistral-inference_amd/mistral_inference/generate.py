"""`generate()` with the reference's contract (generate.py:43-170), minus its host round trips.

Same arguments, same return value: `(tokens, logprobs)` where logprobs hold prompt tokens 1..n-1 then the
generated tokens, `tokens == []` when nothing was generated, sequences keep generating until ALL have hit
`eos_id`, top-p is fixed at 0.8 at the call site (generate.py:126).

What changes is where the bookkeeping runs.  The reference reads one logprob per token with `.item()`
(one device sync per prompt token, generate.py:107,111, and per generated token, :134-136), allocates the
K/V buffers on the host before moving them (:69-78) and rebuilds mask metadata per layer per step.  Here
the cache is allocated in HBM once per call, logprobs are gathered on the device and copied back once at
the end, and a decode step is a single native call with no host metadata; the only per-token sync left
is the EOS test, and only when `eos_id` is given.
"""
import contextlib
from typing import List, Optional, Tuple

import torch

from .cache import BufferCache
from .transformer import Transformer


@torch.inference_mode()
def generate(
    encoded_prompts: List[List[int]],
    model: Transformer,
    images: List[List] = [],
    *,
    max_tokens: int,
    temperature: float,
    chunk_size: Optional[int] = None,
    eos_id: Optional[int] = None,
) -> Tuple[List[List[int]], List[List[float]]]:
    images_torch: List[List[torch.Tensor]] = []
    if images:  # reference generate.py:54-60: one list of [C, H, W] arrays per sample; no chunking with images
        assert chunk_size is None
        images_torch = [[torch.as_tensor(im).to(device=model.device, dtype=model.dtype) for im in images_for_sample]
                        for images_for_sample in images]
    flattened_images: List[torch.Tensor] = sum(images_torch, [])
    model = model.eval()
    B, V = len(encoded_prompts), model.args.vocab_size
    seqlens = [len(x) for x in encoded_prompts]
    dev = model.device

    # K/V rings, straight in device memory (reference generate.py:67-78)
    cache_window = max(seqlens) + max_tokens
    cache = BufferCache(model.n_local_layers, model.args.max_batch_size, cache_window, model.args.n_kv_heads,
                        model.args.head_dim, model.args.sliding_window, device=dev, dtype=model.dtype)
    cache.reset()

    # logprob pieces stay on the device; (sequence, tensor) in emission order
    lp_chunks: List[Tuple[int, torch.Tensor]] = []
    last_token_prelogits: Optional[torch.Tensor] = None

    max_prompt_len = max(seqlens)
    if chunk_size is None:
        chunk_size = max_prompt_len

    # ---- prompt, by chunks (reference generate.py:91-118)
    for s in range(0, max_prompt_len, chunk_size):
        prompt_chunks = [p[s: s + chunk_size] for p in encoded_prompts]
        assert all(len(p) > 0 for p in prompt_chunks)
        flat_ids = sum(prompt_chunks, [])
        if min(flat_ids) < 0 or max(flat_ids) >= V:  # the prompt is host data: raise where nn.Embedding would (transformer.py:193)
            raise IndexError("index out of range in self")
        flat = torch.tensor(flat_ids, device=dev, dtype=torch.long)
        # token i+1 of each chunk is scored by position i
        rows, cols, owners = [], [], []
        offset = 0
        for b, seq in enumerate(prompt_chunks):
            n = len(seq) - 1
            rows += list(range(offset, offset + n))
            cols += seq[1:]
            owners.append((b, n))
            offset += len(seq)
        fused = (getattr(model, "supports_prompt_logprobs", False) and getattr(model, "softmax_fp32", True)
                 and (model.device.type == "cuda" or getattr(model, "prompt_logprobs_any_device", False)))
        if fused:
            # the [T, V] logits are never materialised: the LM head GEMM reduces them to the wanted log-probabilities
            tgt = torch.full((flat.numel(),), -1, dtype=torch.int32)
            if rows:
                tgt[rows] = torch.tensor(cols, dtype=torch.int32)
            lp_rows, last_rows = model.prompt_logprobs(flat, [len(p) for p in prompt_chunks], cache, tgt.to(dev),
                                                       images=flattened_images)
            picked_rows = lp_rows[torch.tensor(rows, device=dev, dtype=torch.long)] if rows else None
        else:
            prelogits = model.forward(flat, seqlens=[len(p) for p in prompt_chunks], cache=cache, images=flattened_images)
            logits = torch.log_softmax(prelogits, dim=-1)
            picked_rows = None
            if rows:
                idx = torch.tensor([rows, cols], device=dev, dtype=torch.long)
                picked_rows = logits[idx[0], idx[1]]
            ends = torch.tensor([len(p) for p in prompt_chunks], device=dev).cumsum(dim=0) - 1
            last_rows = prelogits.index_select(0, ends)

        if last_token_prelogits is not None:
            # first token of this chunk is scored by the previous chunk's last position
            prev = torch.log_softmax(last_token_prelogits, dim=-1)
            firsts = torch.tensor([p[0] for p in prompt_chunks], device=dev, dtype=torch.long)
            picked = prev.gather(1, firsts[:, None])[:, 0]
            for b in range(B):
                lp_chunks.append((b, picked[b: b + 1]))
        if picked_rows is not None:
            o = 0
            for b, n in owners:
                if n:
                    lp_chunks.append((b, picked_rows[o: o + n]))
                o += n
        last_token_prelogits = last_rows
        assert last_token_prelogits.shape == (B, V)

    # ---- decode (reference generate.py:120-140)
    generated: List[torch.Tensor] = []
    gen_lp: List[torch.Tensor] = []
    is_finished = torch.zeros(B, dtype=torch.bool, device=dev)
    assert last_token_prelogits is not None
    fused_greedy = (max_tokens > 0 and hasattr(model, "greedy_session")
                    and (dev.type == "cuda" or (temperature == 0 and getattr(model, "greedy_session_any_device", False)))  # (CPU stand-ins: tests)
                    and (getattr(model, "num_pipeline_ranks", 1) == 1 or getattr(model, "greedy_session_pp", False))
                    and getattr(model, "softmax_fp32", True)
                    and getattr(model, "fused_greedy", True))  # (model.fused_greedy = False: the loop below, for A/B)
    if fused_greedy:
        # One HIP stage - or pipeline stages, each with its own session; the sample then crosses from the last stage to the
        # first as 8 bytes per sequence instead of the reference's [B, vocab] logits broadcast (GreedySession) -: the sample
        # rides on the LM head inside the step (GreedySession) - argmax + log-softmax at temperature
        # 0, the native nucleus draw (csrc/sampling.hip, generate.py:151-170 in one kernel) otherwise - it feeds the next step
        # on the device, and the tokens come back in one copy per CHUNK steps.  Temperature 0: same values as the loop below
        # (token i+1 = first argmax of the logits after token i; its logprob = log_softmax at it).  Temperature > 0: the same
        # distribution as the loop below, drawn from a Philox stream seeded from torch's default generator (torch.manual_seed
        # makes a generation reproducible; torch.multinomial's own stream cannot be reproduced by any other sampler).
        seed = 0
        if temperature > 0:
            from . import _hip
            seed_t = torch.randint(0, 2 ** 62, (1,))
            if getattr(model, "num_pipeline_ranks", 1) > 1:
                # the reference's ranks stay in step only because they share torch's default seed (generate.py:126 on every
                # rank); here only the last stage draws, and rank 0's seed is the generation's seed on every stage
                seed_t = seed_t.to(dev)
                model.pp_comm.broadcast(seed_t, src=0)
            seed = int(seed_t.item())
            next_token, first_lp = _hip.sample_top_p(last_token_prelogits.contiguous(), temperature, 0.8, seed=seed, offset=1 << 63)
        else:
            next_token = sample(last_token_prelogits, temperature=0.0, top_p=0.8)
            lsm = torch.log_softmax(last_token_prelogits, dim=-1)
            first_lp = lsm.gather(1, next_token[:, None])[:, 0]
        if eos_id is not None:
            is_finished = is_finished | (next_token == eos_id)
        if not (eos_id is not None and bool(is_finished.all())):
            generated.append(next_token)
            gen_lp.append(first_lp)
            sess = (model.greedy_session(cache, next_token, temperature=temperature, top_p=0.8, seed=seed) if temperature > 0
                    else model.greedy_session(cache, next_token))
            need = max_tokens - 1            # (the reference's last forward only feeds a sample nobody draws)
            chunk = 32 if eos_id is not None else sess.HIST
            stop = False
            while need > 0 and not stop:
                n = min(need, chunk)
                sess.run(n)
                toks, lps = sess.collect(n)  # [n, B] each; one synchronisation per chunk
                keep = n
                if eos_id is not None:       # generate.py:128-132: stop BEFORE the step at which every sequence has hit EOS
                    fin = is_finished[None, :] | ((toks == eos_id).cumsum(0) > 0)
                    all_fin = fin.all(dim=1)
                    if bool(all_fin.any()):
                        keep = int(torch.nonzero(all_fin)[0, 0])
                        stop = True
                    is_finished = fin[keep - 1] if keep > 0 else is_finished
                for j in range(keep):
                    generated.append(toks[j])
                    gen_lp.append(lps[j])
                need -= n
        max_tokens = 0  # the loop below is done
    # Unfused loop under pipeline parallelism at temperature > 0: every stage samples the same broadcast logits and must draw
    # the same variate whatever state its own CPU generator is in.  ONE seed per generation, agreed here - where the model and
    # its pipeline communicator are known - over `model.pp_comm`; step i then uses (seed, offset i).  `sample()` itself issues
    # no collective: a data-parallel job, or ranks that generate different things, never meet in it (the reference's sample()
    # has none either, generate.py:151-159).
    pp_seed: Optional[int] = None
    if max_tokens > 0 and temperature > 0 and getattr(model, "num_pipeline_ranks", 1) > 1 and hasattr(model, "pp_comm"):
        seed_t = torch.randint(0, 2 ** 62, (1,)).to(dev)
        model.pp_comm.broadcast(seed_t, src=0)
        pp_seed = int(seed_t.item())
    graphed = model.graphed_decode(cache) if hasattr(model, "graphed_decode") else contextlib.nullcontext()
    with graphed:  # decode steps replay a captured hipGraph (single rank; no-op otherwise)
        for step in range(max_tokens):
            next_token = sample(last_token_prelogits, temperature=temperature, top_p=0.8, seed=pp_seed, offset=step)
            if eos_id is not None:
                is_finished = is_finished | (next_token == eos_id)
                if bool(is_finished.all()):  # the one remaining per-token sync, only with an eos_id
                    break
            lsm = torch.log_softmax(last_token_prelogits, dim=-1)
            gen_lp.append(lsm.gather(1, next_token[:, None])[:, 0])
            generated.append(next_token)
            # under the graph the returned tensor is the graph's output buffer (overwritten by the next step);
            # everything derived from it (next token, logprob) is computed before that happens
            last_token_prelogits = model.forward(next_token, seqlens=[1] * B, cache=cache)
            assert last_token_prelogits.shape == (B, V)

    # ---- one copy back
    logprobs: List[List[float]] = [[] for _ in range(B)]
    if lp_chunks:
        flat_lp = torch.cat([t for _, t in lp_chunks]).tolist()
        o = 0
        for b, t in lp_chunks:
            n = t.numel()
            logprobs[b].extend(flat_lp[o: o + n])
            o += n
    generated_tokens: List[List[int]]
    if generated:
        generated_tokens = torch.stack(generated, 1).tolist()
        lp = torch.stack(gen_lp, 1).tolist()
        for b in range(B):
            logprobs[b].extend(lp[b])
    else:
        generated_tokens = []
    # the copies above synchronised the stream: the place to turn device-side flags (an out-of-range token id seen by the
    # embedding kernel, a timed-out wait of the decode engine) into the exceptions the reference would have raised
    backend = getattr(model, "_backend", None)
    if hasattr(backend, "raise_if_flagged"):
        backend.raise_if_flagged()
    return generated_tokens, logprobs


def sample(logits: torch.Tensor, temperature: float, top_p: float, seed: Optional[int] = None, offset: int = 0) -> torch.Tensor:
    """Greedy for temperature 0, else nucleus sampling (reference generate.py:151-159).  fp32 logits on a HIP device: ONE
    native launch (mi_sample_top_p: no sort of the vocabulary, no host sync).

    Seeding contract: `seed=None` draws one 62-bit seed per call from torch's DEFAULT (CPU) generator, i.e. `torch.manual_seed`
    makes a generation reproducible; a seeded CUDA generator (what the reference's torch.multinomial consumes) does not enter.
    A caller that needs several ranks to draw the same token (pipeline stages sampling the same broadcast logits) agrees on a
    seed itself and passes it with a per-step `offset` - `generate()` does, over the model's pipeline communicator; this
    function never communicates."""
    if temperature > 0 and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2:
        from . import _hip
        if seed is None:
            seed, offset = int(torch.randint(0, 2 ** 62, (1,)).item()), 0
        tok, _ = _hip.sample_top_p(logits.contiguous(), temperature, top_p, seed=seed, offset=offset)
        return tok.reshape(-1)
    if temperature > 0:
        probs = torch.softmax(logits / temperature, dim=-1)
        next_token = sample_top_p(probs, top_p)
    else:
        next_token = torch.argmax(logits, dim=-1).unsqueeze(0)
    return next_token.reshape(-1)


def sample_top_p(probs: torch.Tensor, p: float) -> torch.Tensor:
    """Reference generate.py:162-170: keep the smallest prefix of the sorted distribution whose mass
    before the token is <= p, renormalise, draw one."""
    assert 0 <= p <= 1
    sorted_probs, order = torch.sort(probs, dim=-1, descending=True)
    mass_before = torch.cumsum(sorted_probs, dim=-1) - sorted_probs
    sorted_probs = sorted_probs.masked_fill(mass_before > p, 0.0)
    sorted_probs = sorted_probs / sorted_probs.sum(dim=-1, keepdim=True)
    pick = torch.multinomial(sorted_probs, num_samples=1)
    return torch.gather(order, -1, pick)
