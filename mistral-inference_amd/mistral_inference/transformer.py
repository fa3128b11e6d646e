"""`Transformer` with the reference's API (transformer.py:33-338) over the native layer-stack runner.

Drop-in surfaces kept: constructor arguments, `from_folder`, `forward`, `forward_partial`,
`load_state_dict` (rank filtering + "Unexpected key"), the `dtype/device/freqs_cis` properties,
`args`, `n_local_layers`, `layers` (ModuleDict keyed by GLOBAL layer id), `pipeline_rank`,
`num_pipeline_ranks`.  Checkpoint tensor names and layouts are the reference's; weights are used
exactly as loaded (the fused q|k|v and w1|w3 kernels take the separate matrices, nothing is re-packed).

One `forward_partial` = ONE call into libmistral_hip (`mi_forward`): embedding or received activations ->
all local layers -> final norm / LM head, enqueued on the current stream without host synchronisation.
Pipeline parallelism keeps the reference's contract (contiguous layer ranges, transformer.py:94-98;
activations to the next rank, logits broadcast from the last rank) through torch.distributed, whose
"nccl" backend is RCCL over xGMI on ROCm.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import json
import logging
import math
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Any, List, Mapping, Optional, Union

import safetensors.torch
import torch
from torch import nn

from . import _hip
from .args import TransformerArgs
from .cache import BatchMetadata, BufferCache
from .model import ModelBase
from .rope import precompute_freqs_cis
from .transformer_layers import RMSNorm, TransformerBlock
from .vision_encoder import PATCH_MERGE, PatchMerger, VisionLanguageAdapter, VisionTransformer

ROPE_TABLE_LEN = 128_000  # reference transformer.py:116


@dataclass
class SimpleInputMetadata:
    """Positions for the cache=None call (reference transformer.py:21-30)."""

    positions: torch.Tensor

    @staticmethod
    def from_seqlens(seqlens: List[int], device: torch.device) -> "SimpleInputMetadata":
        return SimpleInputMetadata(
            positions=torch.cat([torch.arange(0, s) for s in seqlens]).to(device=device, dtype=torch.long))


@dataclass
class GreedyBuffers:
    """Device buffers of the fused greedy sample (mi_batch_t ABI v4).  `tok` is BOTH the step's input ids and its output:
    the kernels consume the ids before they write the sample, so consecutive steps chain without a copy."""
    tok: torch.Tensor       # int64 [B]
    lp: torch.Tensor        # fp32 [B]
    hist_tok: torch.Tensor  # int64 [hist_len, B]: row (decode steps run on the workspace) % hist_len
    hist_lp: torch.Tensor   # fp32 [hist_len, B]
    temperature: float = 0.0  # > 0: the step's sample is a nucleus draw (mi_batch_t ABI v5) instead of the argmax
    top_p: float = 0.8
    seed: int = 0
    offset: int = 0           # added to the workspace's step counter: the session's draws start at Philox counter 1


def tuned_kernels_take(a: TransformerArgs) -> bool:
    """Shapes the bf16 kernels of `mi_forward` are built for (csrc/api.hip `check_model`); anything else runs on the
    generic kernels (`mi_forward_generic`) instead of raising."""
    E = a.moe.num_experts if a.moe is not None else 0
    k = a.moe.num_experts_per_tok if a.moe is not None else 0
    return (a.head_dim == 128 and a.n_heads % a.n_kv_heads == 0 and a.dim % 8 == 0 and a.hidden_dim % 8 == 0 and a.dim <= 16384
            and E <= 16 and k in (0, 1, 2, 4) and (E == 0 or k * a.hidden_dim * 2 <= 65536))


class HipStackBackend:
    """Runs the local layer stack through `mi_forward`.  The only backend shipped: the product has no
    CPU or eager path (tests may inject a different object to exercise host-side pipeline logic)."""

    def __init__(self) -> None:
        self._plan = None
        self._workspace: Optional[torch.Tensor] = None
        self.generic = False   # set by _build_plan: this model runs through mi_forward_generic
        self.dtype_code = 0

    # -- one-time: pointer tables of the weights -------------------------------------------------
    def _build_plan(self, model: "Transformer"):
        a = model.args
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError(f"mistral_inference needs a HIP device (model is on {dev}); there is no CPU fallback")
        dt = model.dtype
        if dt not in _hip.DTYPE_CODES:
            raise RuntimeError(f"storage dtype {dt}: the HIP kernels take bfloat16 (tuned path), float16 and float32 "
                               "(generic path, csrc/generic.hip)")
        # bf16 models of a shape the tuned kernels take go through mi_forward; everything else - fp16 / fp32 storage
        # (reference transformer.py:303,338 keeps any dtype; its tests build fp32 models, tests/test_generate.py:51) and bf16
        # shapes mi_forward declines with MI_ERR_SHAPE - through mi_forward_generic with the reference's rounding points
        self.generic = dt != torch.bfloat16 or not tuned_kernels_take(a)
        self.dtype_code = _hip.DTYPE_CODES[dt]
        keep = []  # python objects that own memory referenced by raw pointers
        p = lambda t, d=dt: _hip.dev_ptr(t, d)  # noqa: E731
        E = a.moe.num_experts if a.moe is not None else 0
        layers = (_hip.MiLayer * max(1, model.n_local_layers))()
        for j, blk in enumerate(model.layers.values()):
            L = layers[j]
            at = blk.attention
            L.attention_norm, L.ffn_norm = p(blk.attention_norm.weight), p(blk.ffn_norm.weight)
            L.wq, L.wk, L.wv, L.wo = p(at.wq.weight), p(at.wk.weight), p(at.wv.weight), p(at.wo.weight)
            if E:
                ff = blk.feed_forward
                L.gate = p(ff.gate.weight)
                ptrs = []
                for ex in ff.experts:
                    ptrs += [p(ex.w1.weight), p(ex.w2.weight), p(ex.w3.weight)]
                host = _hip.ptr_array(ptrs)
                devtab = torch.tensor(ptrs, dtype=torch.int64, device=dev)
                keep += [host, devtab]
                L.expert_w_host = C.cast(host, C.c_void_p)
                L.expert_w_dev = devtab.data_ptr()
            else:
                ff = blk.feed_forward
                L.w1, L.w2, L.w3 = p(ff.w1.weight), p(ff.w2.weight), p(ff.w3.weight)
        rope = torch.view_as_real(model.freqs_cis).contiguous()
        keep += [layers, rope]
        m = _hip.MiModel()
        m.dim, m.n_heads, m.n_kv_heads, m.head_dim = a.dim, a.n_heads, a.n_kv_heads, a.head_dim
        m.hidden_dim, m.vocab_size, m.n_layers = a.hidden_dim, a.vocab_size, model.n_local_layers
        m.num_experts, m.top_k = E, (a.moe.num_experts_per_tok if E else 0)
        m.norm_eps = a.norm_eps
        m.tok_embeddings = p(model.tok_embeddings.weight) if model.tok_embeddings is not None else None
        m.final_norm = p(model.norm.weight) if model.norm is not None else None
        m.output = p(model.output.weight) if model.output is not None else None
        m.rope_cs, m.rope_len = p(rope, torch.float32), rope.shape[0]
        m.layers = C.cast(layers, C.POINTER(_hip.MiLayer))
        return m, keep

    def plan(self, model: "Transformer"):
        if self._plan is None:
            self._plan = self._build_plan(model)
        return self._plan[0]

    def invalidate(self) -> None:
        self._plan = None
        self._workspace = None

    def raise_if_flagged(self) -> None:
        """Health check at a point where the caller synchronises anyway (end of generate()): an out-of-range token id
        seen by the embedding kernel becomes the reference's IndexError, a timed-out wait of the decode engine a
        RuntimeError."""
        if self._workspace is None:
            return
        st = _hip.decode_engine_status(self._workspace)
        if st["bad_id"]:
            self._workspace[12:16].zero_()
            raise IndexError(f"index out of range in self (token {st['bad_id'] - 1} of a forward call)")
        if st["status"]:
            # the raised word makes every later engine launch on this workspace leave at once: clear it now that it is reported
            _hip.decode_engine_reset(self._workspace)
            what = ("its workgroups were not all resident (GPU shared or CUs masked?); the step wrote nothing"
                    if st["status"] == 0x700 else "a bounded wait timed out; the step's outputs are undefined")
            raise RuntimeError(f"persistent decode engine: status 0x{st['status']:x} - {what}.  "
                               "MI_DECODE_ENGINE=0 selects the launch path.")

    # -- hooks of GreedySession (a test backend provides the same three) ------------------------------
    def session_status(self) -> dict:
        """Control words of the workspace (synchronises): steps run, sticky engine status, ..."""
        return _hip.decode_engine_status(self._workspace)

    _engine_suspended = False  # process-wide: a residency failure switched the engine off until the next session probes again

    def prepare_session(self, model: "Transformer", B: int, cache: BufferCache) -> None:
        if HipStackBackend._engine_suspended:
            # A co-tenant or a CU mask made an earlier step's residency gate fail and the rest of THAT generation ran on the
            # launch path.  Conditions change: a new session forgets the verdict, switches the engine back on and lets the
            # library's census (one probe launch at first use, include/mistral_hip.h) decide again - if it fails, mi_forward
            # takes the launch path on its own; if the gate of a step fails again, collect() suspends the engine again.
            HipStackBackend._engine_suspended = False
            _hip.check(_hip.lib().mi_decode_engine_census(1), "mi_decode_engine_census")
            _hip.set_decode_engine(True)
        self._get_workspace(model, self.plan(model), B, B, max(cache.cache_sizes))  # (a decode step has T == B rows)

    def session_rewind(self, steps: int) -> None:
        """Set the workspace's decode-step counter (status word 5: the row of the greedy history ring and the Philox offset of the
        NEXT step) - GreedySession's lock-step rollback re-runs steps whose samples must land where the first attempt's would."""
        assert self._workspace is not None
        self._workspace[20:24].view(torch.int32).fill_(int(steps) & 0x7FFFFFFF)

    def session_disable_engine(self) -> None:
        """After a raised engine status: clear the word (it poisons the workspace) and take the launch path for the rest of this
        generation; the next session probes the device again (prepare_session)."""
        _hip.decode_engine_reset(self._workspace)
        _hip.set_decode_engine(False)
        HipStackBackend._engine_suspended = True

    def _get_workspace(self, model: "Transformer", m, T: int, B: int, max_w: int) -> torch.Tensor:
        if self.generic:
            need = _hip.lib().mi_workspace_bytes_generic(C.byref(m), T, self.dtype_code)
        else:
            need = _hip.lib().mi_workspace_bytes(C.byref(m), T, B, max_w)
        ws = self._workspace
        if ws is None or ws.numel() < need or ws.device != model.device:
            # grow geometrically; zero-filled because the first 4 KiB are split-KV arrival counters
            ws = torch.zeros(int(need * 1.25) + 4096, dtype=torch.uint8, device=model.device)
            self._workspace = ws
        return ws

    # -- per forward -------------------------------------------------------------------------------
    def run_stack(self, model: "Transformer", h: torch.Tensor, input_ids: Optional[torch.Tensor],
                  meta: BatchMetadata, cache: Optional[BufferCache], logits: Optional[torch.Tensor],
                  greedy: Optional["GreedyBuffers"] = None) -> None:
        m = self.plan(model)
        T, B = h.shape[0], len(meta.seqlens)
        bt = _hip.MiBatch()
        bt.T, bt.B, bt.branch, bt.max_q_len = T, B, meta.branch, meta.max_q_len
        bt.input_ids = _hip.dev_ptr(input_ids, torch.long) if model.tok_embeddings is not None else None
        i32 = torch.int32
        bt.q_start, bt.kv_before = _hip.dev_ptr(meta.q_start, i32), _hip.dev_ptr(meta.kv_before, i32)
        bt.tok_seq, bt.tok_pos = _hip.dev_ptr(meta.tok_seq, i32), _hip.dev_ptr(meta.tok_pos, i32)
        max_w = 1
        if cache is not None:
            if cache.n_layers and cache.cache_k[0].dtype != model.dtype:
                raise RuntimeError(f"cache dtype {cache.cache_k[0].dtype} != model dtype {model.dtype} (BufferCache.to(device, dtype))")
            ks, vs, ws, bt.kv_layout = cache.pointer_tables()
            bt.cache_k, bt.cache_v = C.cast(ks, C.POINTER(C.c_void_p)), C.cast(vs, C.POINTER(C.c_void_p))
            bt.cache_sizes = C.cast(ws, C.POINTER(C.c_int32))
            bt.kv_seqlens = _hip.dev_ptr(cache.kv_seqlens, torch.long)
            max_w = max(cache.cache_sizes)
        bt.h = _hip.dev_ptr(h, model.dtype)
        bt.logits = _hip.dev_ptr(logits, torch.float32)
        if greedy is not None:  # ABI v4: sample fused behind the LM head (generate.py:124,134-136 at temperature 0)
            bt.greedy_token, bt.greedy_logprob = _hip.dev_ptr(greedy.tok, torch.long), _hip.dev_ptr(greedy.lp, torch.float32)
            bt.hist_token, bt.hist_logprob = _hip.dev_ptr(greedy.hist_tok, torch.long), _hip.dev_ptr(greedy.hist_lp, torch.float32)
            bt.hist_len = greedy.hist_tok.shape[0]
            bt.sample_temperature, bt.sample_top_p = float(greedy.temperature), float(greedy.top_p)
            bt.sample_seed = int(greedy.seed) & (2 ** 64 - 1)
            bt.sample_offset = int(greedy.offset) & (2 ** 64 - 1)
        wsb = self._get_workspace(model, m, T, B, max_w)
        bt.workspace, bt.workspace_bytes = wsb.data_ptr(), wsb.numel()
        if self.generic:
            _hip.check(_hip.lib().mi_forward_generic(C.byref(m), C.byref(bt), self.dtype_code, _hip.stream_ptr(h.device)),
                       "mi_forward_generic")
        else:
            _hip.check(_hip.lib().mi_forward(C.byref(m), C.byref(bt), _hip.stream_ptr(h.device)), "mi_forward")


class Transformer(ModelBase):
    greedy_session_pp = True  # generate(): the fused sampling session also runs across pipeline stages (GreedySession)

    def __init__(self, args: TransformerArgs, pipeline_rank: int = 0, num_pipeline_ranks: int = 1,
                 softmax_fp32: bool = True, backend: Optional[Any] = None):
        super().__init__()
        self.args = args
        self.vocab_size = args.vocab_size
        self.n_layers = args.n_layers
        self._precomputed_freqs_cis: Optional[torch.Tensor] = None
        assert self.vocab_size > 0
        assert pipeline_rank < num_pipeline_ranks, (pipeline_rank, num_pipeline_ranks)
        self.pipeline_rank = pipeline_rank
        self.num_pipeline_ranks = num_pipeline_ranks
        self.softmax_fp32 = softmax_fp32
        if args.lora is not None:
            raise NotImplementedError("un-merged LoRA is outside the hot path; merge the adapter into the weights")

        # Rank-specific modules (reference transformer.py:52-79)
        self.tok_embeddings: Optional[nn.Embedding] = None
        self.norm: Optional[RMSNorm] = None
        self.output: Optional[nn.Linear] = None
        self.vision_encoder: Optional[VisionTransformer] = None
        self.vision_language_adapter: Optional[VisionLanguageAdapter] = None
        if pipeline_rank == 0:
            self.tok_embeddings = nn.Embedding(args.vocab_size, args.dim)
            if args.vision_encoder is not None:  # Pixtral tower lives on the first stage (reference transformer.py:59-76)
                ve = args.vision_encoder
                self.vision_encoder = VisionTransformer(ve)
                self.vision_language_adapter = VisionLanguageAdapter(ve.hidden_size, args.dim, ve.adapter_bias)
                if ve.add_pre_mm_projector_layer_norm:
                    self.pre_mm_projector_norm = RMSNorm(ve.hidden_size, eps=1e-5)
                if ve.mm_projector_id == PATCH_MERGE:
                    self.patch_merger = PatchMerger(vision_encoder_dim=ve.hidden_size,
                                                    spatial_merge_size=ve.spatial_merge_size)
        if pipeline_rank == num_pipeline_ranks - 1:
            self.norm = RMSNorm(args.dim, eps=args.norm_eps)
            self.output = nn.Linear(args.dim, args.vocab_size, bias=False)
        # Only this rank's contiguous layer range is ever constructed (the reference builds all n_layers
        # blocks and throws the rest away, transformer.py:80-98); keys stay GLOBAL layer ids.
        per_rank = math.ceil(self.n_layers / self.num_pipeline_ranks)
        first = self.pipeline_rank * per_rank
        last = min(self.n_layers, first + per_rank)
        self.layers = nn.ModuleDict({
            str(i): TransformerBlock(dim=args.dim, hidden_dim=args.hidden_dim, n_heads=args.n_heads,
                                     n_kv_heads=args.n_kv_heads, head_dim=args.head_dim, norm_eps=args.norm_eps,
                                     lora=args.lora, moe=args.moe)
            for i in range(first, last)})
        self.n_local_layers = len(self.layers)
        self._backend = backend if backend is not None else HipStackBackend()
        self._graphed: Optional[dict] = None  # state of an active graphed_decode() context
        self._pp_comm: Optional[Any] = None   # pipeline transport (distributed.pipeline_comm), created at first use

    # ---- properties ----------------------------------------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def freqs_cis(self) -> torch.Tensor:
        """complex64 [128000, head_dim/2] as in the reference (transformer.py:107-120); the kernels read its
        real view."""
        if self._precomputed_freqs_cis is None:
            theta = self.args.rope_theta or 1000000.0
            self._precomputed_freqs_cis = precompute_freqs_cis(self.args.head_dim, ROPE_TABLE_LEN, theta)
        if self._precomputed_freqs_cis.device != self.device:
            self._precomputed_freqs_cis = self._precomputed_freqs_cis.to(device=self.device)
        return self._precomputed_freqs_cis

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / dtype casts move the weights: rebuild pointer tables
        out = super()._apply(fn, *a, **k)
        self._weights_changed()
        return out

    # ---- pipeline transport ----------------------------------------------------------------------
    @property
    def pp_comm(self):
        """RCCL communicator through the C ABI on a GPU (stream-ordered, capturable), torch.distributed otherwise."""
        if self._pp_comm is None:
            from .distributed import pipeline_comm
            self._pp_comm = pipeline_comm(self.device)
        return self._pp_comm

    # ---- forward -------------------------------------------------------------------------------
    def _nocache_metadata(self, seqlens: List[int]) -> BatchMetadata:
        T = sum(seqlens)
        pos = [i for s in seqlens for i in range(s)]
        blob = torch.tensor([0, T] + [0] + [0] * T + pos, dtype=torch.int32).to(self.device)
        return BatchMetadata(_hip.BRANCH_NOCACHE, [T], T, blob[:2], blob[2:3], blob[3:3 + T], blob[3 + T:])

    def embed_vision_language_features(self, input_ids: torch.Tensor, images: List[torch.Tensor]) -> torch.Tensor:
        """Token embeddings with the image-token rows replaced by the adapter's image features, in order (reference
        transformer.py:122-161)."""
        assert self.tok_embeddings is not None
        assert self.vision_encoder is not None
        assert self.vision_language_adapter is not None
        ve = self.args.vision_encoder
        assert ve is not None
        dev = self.device
        input_ids = input_ids.to(dev)
        image_locations = input_ids == ve.image_token_id
        text_locations = ~image_locations
        image_features = self.vision_encoder(images)
        if ve.add_pre_mm_projector_layer_norm:
            image_features = self.pre_mm_projector_norm(image_features)
        if ve.mm_projector_id == PATCH_MERGE:
            img_patch_dims = [(img.shape[1] // ve.patch_size, img.shape[2] // ve.patch_size) for img in images]
            image_features = self.patch_merger(image_features, image_sizes=img_patch_dims)
        image_features = self.vision_language_adapter(image_features)
        N_txt, (N_img, D_img) = int(text_locations.sum()), image_features.shape
        seq_len = input_ids.shape[0]
        assert self.args.dim == D_img, f"Text features dim {self.args.dim} should be equal to image features dim {D_img}"
        assert seq_len == N_txt + N_img, (
            f"seq_len {seq_len} should be equal to N_txt + N_img {(N_txt, N_img, int(image_locations.sum()))}")
        combined = torch.empty((seq_len, D_img), dtype=self.dtype, device=dev)
        if N_txt:
            combined[text_locations, :] = _hip.embedding(self.tok_embeddings.weight, input_ids[text_locations])
        combined[image_locations, :] = image_features
        return combined

    def _run(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[BufferCache],
             want_logits: bool, images: Optional[List[torch.Tensor]] = None):
        assert len(seqlens) <= self.args.max_batch_size, (
            f"Max batch size is {self.args.max_batch_size}, got batch size of {len(seqlens)}")
        (num_toks,) = input_ids.shape
        assert sum(seqlens) == num_toks, (sum(seqlens), num_toks)
        if self.pipeline_rank == 0:
            _hip.check_ids_on_host(input_ids, self.vocab_size)  # device-resident ids: flagged by the embedding kernel
        meta = cache.batch_metadata(seqlens) if cache is not None else self._nocache_metadata(seqlens)
        # the kernels index the rotary table by position without a bound check (the reference's gather would raise)
        top = max((p + s for p, s in zip(cache._seen, seqlens)), default=0) if cache is not None else max(seqlens)
        if top > ROPE_TABLE_LEN:
            raise IndexError(f"position {top - 1} is beyond the {ROPE_TABLE_LEN}-entry rotary table (reference transformer.py:116)")
        dev = self.device
        multimodal = self.pipeline_rank == 0 and self.vision_encoder is not None and bool(images)
        if multimodal:  # reference transformer.py:190-191
            h = self.embed_vision_language_features(input_ids, images)
        else:
            h = torch.empty((num_toks, self.args.dim), device=dev, dtype=self.dtype)
        if self.pipeline_rank > 0:
            self.pp_comm.recv(h, src=self.pipeline_rank - 1)
        last = self.pipeline_rank == self.num_pipeline_ranks - 1
        logits = None
        if want_logits and last:
            logits = torch.empty((num_toks, self.vocab_size), device=dev, dtype=torch.float32)
        ids = input_ids.to(device=dev, dtype=torch.long) if (self.pipeline_rank == 0 and not multimodal) else None
        self._backend.run_stack(self, h, ids, meta, cache, logits)
        if cache is not None:
            if meta.branch == _hip.BRANCH_DECODE:
                cache.advance_host(seqlens)   # device kv_seqlens already advanced by the decode-prep kernel
            else:
                cache.update_seqlens(seqlens)
        if not last:
            self.pp_comm.send(h, dst=self.pipeline_rank + 1)
        return h, logits

    def forward_partial(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[BufferCache] = None,
                        images: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """Local forward pass (reference transformer.py:163-219): the activations of this stage's last layer,
        RMS-normalised on the last stage."""
        h, _ = self._run(input_ids, seqlens, cache, want_logits=False, images=images)
        return h

    # ---- decode step as a hipGraph ----------------------------------------------------------------
    @contextlib.contextmanager
    def graphed_decode(self, cache: BufferCache):
        """Inside this context, decode-branch `forward()` calls on `cache` (every sequence adds exactly one token)
        are replayed from a captured hipGraph instead of being enqueued launch by launch.

        A decode step is ~165 dependent kernel launches and needs nothing from the host (positions and ring slots
        are derived on the device from `cache.kv_seqlens`), so the whole step is captured once - on the second
        decode call, after one eager warm-up step - and replayed afterwards; only the token ids are copied into the
        graph's input buffer.  The returned logits tensor is the graph's output buffer: it is overwritten by the
        next call, which is how `generate()` uses it (it reduces the logits to a token and a logprob immediately).
        Under pipeline parallelism the RCCL transfers are captured with the step (torch.distributed transports - the
        gloo CPU tests - run eagerly: the context is a no-op there).
        """
        from .distributed import RcclComm
        usable = (self.device.type == "cuda" and isinstance(self._backend, HipStackBackend)
                  and (self.num_pipeline_ranks == 1 or isinstance(self.pp_comm, RcclComm)))
        self._graphed = {"cache": cache, "graph": None, "warm": 0} if usable else None
        try:
            yield self
        finally:
            self._graphed = None

    # ---- greedy decoding with the sample fused into the step ------------------------------------------
    def greedy_session(self, cache: BufferCache, first_tokens: torch.Tensor, graph: bool = True, temperature: float = 0.0,
                       top_p: float = 0.8, seed: int = 0) -> "GreedySession":
        """Decode loop of `generate()` (reference generate.py:120-140) with nothing but the model's own launches per token:
        see GreedySession.  temperature 0: argmax; temperature > 0: the nucleus draw of generate.py:151-170 on the device."""
        return GreedySession(self, cache, first_tokens, graph, temperature=temperature, top_p=top_p, seed=seed)

    def _logits(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[BufferCache],
                images: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """One forward up to the logits every rank returns (reference transformer.py:221-242)."""
        h, logits = self._run(input_ids, seqlens, cache, want_logits=True, images=images)
        if self.num_pipeline_ranks > 1:
            # every rank receives the logits in the model dtype, as the reference does (transformer.py:229-237);
            # the kernel's fp32 logits are bf16-representable, so the narrowing is exact
            outs = logits.to(self.dtype) if logits is not None else torch.empty(
                h.shape[0], self.vocab_size, device=h.device, dtype=h.dtype)
            self.pp_comm.broadcast(outs, src=self.num_pipeline_ranks - 1)
            return outs.float() if self.softmax_fp32 else outs
        assert logits is not None
        return logits if self.softmax_fp32 else logits.to(self.dtype)

    def _graphed_step(self, input_ids: torch.Tensor, seqlens: List[int], cache: BufferCache, st: dict) -> torch.Tensor:
        if max(cache._seen) + 1 > ROPE_TABLE_LEN:  # same bound as _run (a replayed step never passes through it)
            raise IndexError(f"position {max(cache._seen)} is beyond the {ROPE_TABLE_LEN}-entry rotary table")
        if st["graph"] is None:
            if st["warm"] < 1:  # eager step: sizes the workspace and the cache's device metadata before capture
                st["warm"] += 1
                return self._logits(input_ids, seqlens, cache)
            st["ids"] = input_ids.to(device=self.device, dtype=torch.long).clone()
            st["B"] = len(seqlens)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            seen = list(cache._seen)
            captured, why = True, None
            try:
                with torch.cuda.graph(graph):
                    # under pipeline parallelism the stage-to-stage ncclRecv / ncclSend and the logits broadcast are graph
                    # nodes too (RCCL enqueues on the capturing stream): a replay needs no Python per token on any rank
                    st["out"] = self._logits(st["ids"], seqlens, cache)
            except RuntimeError as e:
                captured, why = False, e
            if self.num_pipeline_ranks > 1 and torch.distributed.is_initialized():
                # every stage replays or every stage runs eagerly: a rank that could not capture while its neighbours
                # replay would post its sends / receives in a different order (agreed over the bootstrap process group)
                flag = torch.tensor([1 if captured else 0], device=self.device, dtype=torch.int32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                captured = bool(int(flag.item()))
            if not captured:
                # a runtime that refuses to capture (an RCCL build without graph support, a debug allocator ...): this
                # context degrades to launch-by-launch steps instead of failing the generation
                logging.warning("decode step not graph-capturable on every stage (%s): continuing eagerly", why)
                cache._seen = seen
                self._graphed = None
                return self._logits(input_ids, seqlens, cache)
            cache._seen = seen  # capture enqueues nothing: the step itself is the first replay below
            st["graph"] = graph
        else:
            st["ids"].copy_(input_ids)
        st["graph"].replay()
        cache.advance_host(seqlens)
        return st["out"]

    def forward(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[BufferCache] = None,
                images: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """Logits [T, vocab] (reference transformer.py:221-242): fp32 unless softmax_fp32=False."""
        st = self._graphed
        if (st is not None and cache is st["cache"] and cache._seen is not None and cache._seen[0] > 0
                and all(s == 1 for s in seqlens) and len(seqlens) == len(cache._seen)
                and (st["graph"] is None or len(seqlens) == st["B"])):
            return self._graphed_step(input_ids, seqlens, cache, st)
        return self._logits(input_ids, seqlens, cache, images)

    @property
    def supports_prompt_logprobs(self) -> bool:
        """generate() reduces a prompt chunk's logits inside the LM head GEMM (prompt_logprobs) on the tuned bf16 path; models
        on the generic kernels return [T, vocab] logits from forward() and generate() takes its log-softmax."""
        be = self._backend
        if isinstance(be, HipStackBackend):
            if self.device.type != "cuda":
                return False
            be.plan(self)
            return not be.generic
        return bool(getattr(be, "supports_prompt_logprobs", True))

    def prompt_logprobs(self, input_ids: torch.Tensor, seqlens: List[int], cache: Optional[BufferCache],
                        targets: torch.Tensor, images: Optional[List[torch.Tensor]] = None):
        """What `generate()` takes from a prompt chunk's logits (reference generate.py:101-118), without the
        [T, vocab] fp32 tensor `forward` contractually returns: (a) fp32 [T]
        log_softmax(logits)[t, targets[t]] (targets[t] < 0: ignored row) from ONE pass over the LM head with the
        log-softmax reduction in the GEMM epilogue, (b) the fp32 logits of each sequence's LAST row, [B, vocab].
        fp32-softmax models only (callers fall back to `forward` otherwise).  Under pipeline parallelism the last
        rank computes both and broadcasts T + B * vocab floats instead of the reference's [T, vocab] logits."""
        assert self.softmax_fp32
        h, _ = self._run(input_ids, seqlens, cache, want_logits=False, images=images)  # final RMSNorm on the last rank
        T, B, dev = h.shape[0], len(seqlens), h.device
        last_rank = self.pipeline_rank == self.num_pipeline_ranks - 1
        targets = targets.to(device=dev, dtype=torch.int32)
        if self.num_pipeline_ranks > 1:
            # only rank 0 is guaranteed to hold the real prompt (the reference's other ranks get placeholders of the same
            # length, main.py:147-160): its target ids travel to the rank that owns the LM head
            self.pp_comm.broadcast(targets, src=0)
        if last_rank:
            assert self.output is not None
            lp = _hip.lm_head_logprobs(h, self.output.weight, targets)
            ends = torch.tensor(seqlens, device=dev).cumsum(dim=0) - 1
            last = _hip.linear(h.index_select(0, ends).contiguous(), (self.output.weight,), _hip.EPI_LOGITS)
        else:
            lp = torch.empty(T, dtype=torch.float32, device=dev)
            last = torch.empty((B, self.vocab_size), dtype=torch.float32, device=dev)
        if self.num_pipeline_ranks > 1:
            self.pp_comm.broadcast(lp, src=self.num_pipeline_ranks - 1)
            self.pp_comm.broadcast(last, src=self.num_pipeline_ranks - 1)
        return lp, last

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True, assign: bool = False) -> None:
        """Keep only this rank's tensors (reference transformer.py:244-295): embeddings on rank 0, norm/output
        on the last rank, layers by global id; any other key raises ValueError("Unexpected key ...")."""
        mine, skipped = {}, set()
        last = self.pipeline_rank == self.num_pipeline_ranks - 1
        for k, v in state_dict.items():
            if k.startswith("tok_embeddings"):
                keep = self.pipeline_rank == 0
            elif k.startswith("norm") or k.startswith("output"):
                keep = last
            elif k.startswith("layers"):
                keep = k.split(".")[1] in self.layers
            elif any(k.startswith(p) for p in ("vision_encoder", "vision_language_adapter", "patch_merger",
                                               "pre_mm_projector_norm")):
                keep = self.pipeline_rank == 0
            else:
                raise ValueError(f"Unexpected key {k}")
            if keep:
                mine[k] = v
            else:
                logging.debug("Skipping parameter %s at pipeline rank %d", k, self.pipeline_rank)
                skipped.add(k)
        assert set(state_dict.keys()) == skipped.union(set(mine.keys()))
        super().load_state_dict(mine, strict=strict, assign=assign)
        self._weights_changed()

    # ---- LoRA (reference lora.py:92-139): adapters are MERGED into the frozen weights at load time -------------
    def load_lora(self, lora_path: Union[Path, str], scaling: float = 2.0) -> None:
        """Loads a LoRA checkpoint (safetensors with `<linear>.lora_A.weight` / `.lora_B.weight` keys) and folds it
        into the weights: W <- W + (B @ A) * scaling, every nn.Linear of this rank's layers except `output`."""
        lora_path = Path(lora_path)
        assert lora_path.is_file(), f"{lora_path} does not exist or is not a file"
        self._load_lora_state_dict(safetensors.torch.load_file(str(lora_path)), scaling=scaling)

    def _load_lora_state_dict(self, lora_state_dict: Mapping[str, torch.Tensor], scaling: float = 2.0) -> None:
        lora_dtypes = set(p.dtype for p in lora_state_dict.values())
        assert len(lora_dtypes) == 1, (
            f"LoRA weights have multiple different dtypes {lora_dtypes}. All weights need to have the same dtype")
        lora_dtype = lora_dtypes.pop()
        assert lora_dtype == self.dtype, f"LoRA weights dtype differs from model's dtype {lora_dtype} != {self.dtype}"
        assert all("lora" in key for key in lora_state_dict.keys())
        if self.dtype != torch.bfloat16 or self.device.type != "cuda":
            raise RuntimeError("load_lora: the merge runs on the GPU in bf16 (model must be on the device)")
        logging.info("Loading and merging LoRA weights...")
        sd = {k: v.to(self.device) for k, v in lora_state_dict.items()}
        with torch.no_grad():
            for name, module in self.named_modules():
                if not isinstance(module, nn.Linear) or name == "output":
                    continue
                if name.split(".")[1] not in self.layers:
                    logging.debug("Skipping parameter %s at pipeline rank %d", name, self.pipeline_rank)
                    continue
                if name + ".lora_B.weight" not in sd:
                    continue
                a, b = sd[name + ".lora_A.weight"], sd[name + ".lora_B.weight"]  # [r, in], [out, r]
                r = a.shape[0]
                pad = (-r) % 8  # the GEMM wants K % 8 == 0: zero columns add nothing
                at = torch.nn.functional.pad(a.t(), (0, pad)).contiguous()  # [in, r + pad]
                bp = torch.nn.functional.pad(b, (0, pad)).contiguous()      # [out, r + pad]
                delta = _hip.linear(bp, (at,), _hip.EPI_STORE)              # bf16(B @ A), fp32 accumulation
                module.weight.copy_(module.weight + delta * scaling)        # same rounding points as lora.py:131-135
        if hasattr(self._backend, "invalidate"):
            self._backend.invalidate()

    @staticmethod
    def from_folder(folder: Union[Path, str], max_batch_size: int = 1, num_pipeline_ranks: int = 1,
                    device: Union[torch.device, str] = "cuda", dtype: Optional[torch.dtype] = None,
                    softmax_fp32: bool = True, backend: Optional[Any] = None) -> "Transformer":
        """params.json + exactly one of consolidated.00.pth / consolidated.safetensors (reference
        transformer.py:297-338).  With safetensors only this rank's tensors are read, straight to `device`
        (the reference loads the whole file to the host on every rank)."""
        folder = Path(folder)
        with open(folder / "params.json", "r") as f:
            model_args = TransformerArgs.from_dict(json.load(f))
        model_args.max_batch_size = max_batch_size
        pipeline_rank = torch.distributed.get_rank() if num_pipeline_ranks > 1 else 0
        with torch.device("meta"):
            model = Transformer(model_args, pipeline_rank=pipeline_rank, num_pipeline_ranks=num_pipeline_ranks,
                                softmax_fp32=softmax_fp32, backend=backend)
        pt_file = folder / "consolidated.00.pth"
        st_file = folder / "consolidated.safetensors"
        assert pt_file.exists() or st_file.exists(), f"Make sure either {pt_file} or {st_file} exists"
        assert not (pt_file.exists() and st_file.exists()), f"Both {pt_file} and {st_file} cannot exist"
        if pt_file.exists():
            loaded = torch.load(str(pt_file), mmap=True)
            model.load_state_dict(loaded, assign=True, strict=True)
            return model.to(device=device, dtype=dtype)
        wanted = set(model.state_dict().keys())
        loaded = {}
        with safetensors.safe_open(str(st_file), framework="pt", device=str(device)) as f:
            for k in f.keys():
                if k in wanted:
                    loaded[k] = f.get_tensor(k)
                else:
                    model._check_foreign_key(k)
        missing = wanted - set(loaded)
        assert not missing, f"checkpoint is missing {sorted(missing)[:4]}..."
        nn.Module.load_state_dict(model, loaded, strict=True, assign=True)
        model._weights_changed()
        return model.to(device=device, dtype=dtype)

    def _weights_changed(self) -> None:
        """Pointer tables / derived weight images are rebuilt lazily after any (re)load."""
        if hasattr(self._backend, "invalidate"):
            self._backend.invalidate()
        if self.vision_language_adapter is not None:
            self.vision_language_adapter.invalidate()
        if self.vision_encoder is not None:
            self.vision_encoder._conv_w = None

    def _check_foreign_key(self, k: str) -> None:
        """A key this rank does not own must still be a known kind (reference raises on anything else)."""
        known = ("tok_embeddings", "norm", "output", "layers", "vision_encoder", "vision_language_adapter", "patch_merger",
                 "pre_mm_projector_norm")
        if not any(k.startswith(p) for p in known):
            raise ValueError(f"Unexpected key {k}")


class GreedySession:
    """`next_token = argmax(logits); logprob = log_softmax(logits)[next_token]; logits = model.forward(next_token)`
    (reference generate.py:124-140, temperature 0) as ONE native call per token with no torch launch in between.

    * the LM head's epilogue produces the sample (persistent engine: in the same launch; launch path: one more small
      kernel) and writes it (a) into the id buffer the NEXT step reads - `mi_batch_t.input_ids` may alias
      `greedy_token` - and (b) into a history ring on the device;
    * on the launch path a step (~165 launches) is captured once in a hipGraph and replayed; on the persistent engine a step
      is a single kernel and is launched plainly (a one-kernel graph costs more per replay than a queued launch, round 6);
    * `run(n)` enqueues n steps without touching the host again; `collect(n)` reads the n samples back in one copy,
      which is also where a device-side failure is noticed: an engine step whose residency census failed (status
      0x700, include/mistral_hip.h) wrote nothing, so the missing steps are re-run on the launch path.

    The [B, vocab] fp32 logits of every step are still produced (`self.logits`): the work of `forward()` is unchanged,
    only what is done with its result moved onto the device.

    temperature > 0 (`mistral-chat`'s default, reference main.py:105 + generate.py:126,151-170): the same session with the
    argmax replaced by the native nucleus draw (csrc/sampling.hip: one more small kernel behind the LM head inside the same
    captured step, Philox variate keyed by the seed and the workspace's step counter) - still one native call per token.

    PIPELINE STAGES (num_pipeline_ranks > 1; reference transformer.py:195-196,213-214,236-237 per token): every stage runs
    this session on its own layer range.  Per token: stage r receives the [B, dim] activations of stage r - 1, runs ONE
    native call, sends them on; the LAST stage's call ends in the sample, and that sample - 8 bytes per sequence - goes
    back to stage 0 as the next step's input id.  The reference broadcasts the [B, vocab] logits to every rank per token
    (64 KiB at vocab 32768) and lets every rank sample; here nothing of vocab size crosses a link at decode, and the other
    ranks learn the tokens (generate() returns them on every rank) from ONE broadcast of the history per collect().
    With the C-ABI RCCL transport (MI_PP_TRANSPORT=rccl) the hops are stream-ordered and part of the captured step."""

    HIST = 1024
    # decode steps per hipGraph launch.  Measured (profiles/EXPERIMENTS.md): 8 steps per graph close the ~9 us gap between two
    # graph launches, and the kernels then run ~10 us longer each (their ramp-up is no longer hidden in the gap): no gain -> 1
    GRAPH_STEPS = int(os.environ.get("MI_GRAPH_STEPS", "1"))

    def __init__(self, model: "Transformer", cache: BufferCache, first_tokens: torch.Tensor, graph: bool = True,
                 temperature: float = 0.0, top_p: float = 0.8, seed: int = 0):
        be = model._backend
        assert hasattr(be, "session_status"), "the fused sampling step needs a backend with a step counter (HipStackBackend)"
        dev = model.device
        self.model, self.cache = model, cache
        self.rank, self.world = model.pipeline_rank, model.num_pipeline_ranks
        self.is_last = self.rank == self.world - 1
        B = int(first_tokens.numel())
        self.B = B
        assert cache._seen is not None and len(cache._seen) == B and cache._seen[0] > 0, "prefill the cache first"
        self.buf = GreedyBuffers(tok=first_tokens.to(device=dev, dtype=torch.long).reshape(B).clone(),
                                 lp=torch.zeros(B, dtype=torch.float32, device=dev),
                                 hist_tok=torch.zeros((self.HIST, B), dtype=torch.long, device=dev),
                                 hist_lp=torch.zeros((self.HIST, B), dtype=torch.float32, device=dev),
                                 temperature=float(temperature), top_p=float(top_p), seed=int(seed))
        self.logits = torch.empty((B, model.vocab_size), dtype=torch.float32, device=dev) if self.is_last else None
        self.h = torch.empty((B, model.args.dim), dtype=model.dtype, device=dev)
        from .distributed import RcclComm
        # hops through torch.distributed are not capturable: such stages step launch by launch (ONE launch on the engine)
        self._use_graph = (graph and dev.type == "cuda" and isinstance(be, HipStackBackend)
                           and (self.world == 1 or isinstance(model.pp_comm, RcclComm)))
        self._graphs: dict = {}            # steps per graph -> captured hipGraph
        self._warm = False
        self._base: Optional[int] = None   # value of the workspace's step counter when this session began
        self._pending = 0                  # steps enqueued and not yet collected
        self._n_collected = 0
        self._first = self.buf.tok.clone()  # input of the session's step 0 (a rollback to step 0 needs it again)

    # -- one step, enqueued launch by launch
    def _step_eager(self) -> None:
        m, cache = self.model, self.cache
        meta = cache.batch_metadata([1] * self.B)
        assert meta.branch == _hip.BRANCH_DECODE
        if self.rank > 0:
            m.pp_comm.recv(self.h, src=self.rank - 1)
        ids = self.buf.tok if self.rank == 0 else None
        m._backend.run_stack(m, self.h, ids, meta, cache, self.logits, greedy=self.buf if self.is_last else None)
        if not self.is_last:
            m.pp_comm.send(self.h, dst=self.rank + 1)
        if self.world > 1:
            # the sample travels from the last stage to the first: 8 bytes per sequence (the reference: [B, vocab] logits to
            # every rank, transformer.py:236-237).  Stage 0 receives it into the id buffer its NEXT step reads.
            if self.is_last:
                m.pp_comm.send(self.buf.tok, dst=0)
            elif self.rank == 0:
                m.pp_comm.recv(self.buf.tok, src=self.world - 1)

    def _steps_now(self) -> int:
        return self.model._backend.session_status()["steps"]

    def _one_step(self) -> None:
        m, cache = self.model, self.cache
        if not self._warm:  # first step eagerly: sizes the workspace, runs the engine's one-time residency census
            # size the workspace exactly as run_stack will BEFORE reading its step counter: a re-allocation inside the first
            # step would restart the counter at zero and collect() would index the wrong history rows
            m._backend.prepare_session(m, self.B, cache)
            st0 = m._backend.session_status()
            self._base = st0["steps"]
            self.buf.offset = -self._base  # (mod 2^64) a generation's variates do not depend on what ran on the workspace before
            self._step_eager()
            self._warm = True
            if self._use_graph and os.environ.get("MI_ENGINE_GRAPH", "0") != "1" and not getattr(self, "engine_graph", False):
                # Did this step run on the persistent engine?  Then a step is ONE kernel and plain launches, which the host
                # enqueues ~50x faster than the GPU retires them, run back to back - whereas replaying a one-kernel hipGraph per
                # step costs ~28 us more per token than a queued plain launch (round 6: bench.py roofline.other_loops_us,
                # 2624 vs 2596 us per step = 1.1 %).  The hipGraph stays for the launch path (~165 launches per step).
                # One synchronisation per session; MI_ENGINE_GRAPH=1 keeps the replayed form (A/B).
                if m._backend.session_status()["engine_launches"] > st0["engine_launches"]:
                    self._use_graph = False
            return
        if self._use_graph:
            g = self._captured(1)
            if g is not None:
                g.replay()
                return
        self._step_eager()

    def _captured(self, steps: int):
        """hipGraph of `steps` consecutive decode steps (the sample of one is the input of the next on the device, so a
        graph may hold any number of them).  Between two graph launches the GPU idles for a few microseconds that it does
        not between two kernels of one graph: the loop is replayed GRAPH_STEPS steps at a time."""
        g = self._graphs.get(steps)
        if g is None:
            torch.cuda.synchronize(self.model.device)
            g = torch.cuda.CUDAGraph()
            ok, why = True, None
            try:
                with torch.cuda.graph(g):  # (capture enqueues nothing: the steps themselves are the replays)
                    for _ in range(steps):
                        self._step_eager()
            except RuntimeError as e:  # a runtime that refuses to capture: launch by launch from here on
                ok, why = False, e
            if self.world > 1 and torch.distributed.is_initialized():
                # every stage replays or every stage steps eagerly (a stage that could not capture while its neighbours
                # replay would post its hops in a different order): agreed over the bootstrap process group
                flag = torch.tensor([1 if ok else 0], device=self.model.device, dtype=torch.int32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = bool(int(flag.item()))
            if not ok:
                logging.warning("decode step not graph-capturable on every stage (%s): continuing eagerly", why)
                self._use_graph = False
                return None
            self._graphs[steps] = g
        return g

    def run(self, n: int) -> None:
        """Enqueue n decode steps (no host synchronisation once the steps have been captured)."""
        cache = self.cache
        if max(cache._seen) + n > ROPE_TABLE_LEN:
            raise IndexError(f"position {max(cache._seen) + n - 1} is beyond the {ROPE_TABLE_LEN}-entry rotary table")
        assert self._pending + n <= self.HIST, "collect() before the history ring wraps"
        left = n
        while left > 0:
            k = 1
            if self._warm and self._use_graph and left >= self.GRAPH_STEPS > 1:
                g = self._captured(self.GRAPH_STEPS)
                if g is not None:
                    g.replay()
                    k = self.GRAPH_STEPS
            if k == 1:
                self._one_step()
            cache.advance_host([k] * self.B)
            self._pending += k
            left -= k

    def collect(self, n: Optional[int] = None):
        """(tokens int64 [n, B], logprobs fp32 [n, B]) of the n oldest uncollected steps, on the host side of one
        synchronisation; verifies that the device really ran them (and re-runs what an engine failure skipped).  Under
        pipeline parallelism the last stage's history is broadcast to every stage here (once per collect, not per token)."""
        n = self._pending if n is None else n
        assert 0 < n <= self._pending
        m = self.model
        be = m._backend
        st = be.session_status()  # synchronises
        assert self._base is not None
        done_total = st["steps"] - self._base      # steps the device completed since the session began
        issued_total = self._issued()
        status = st["status"]
        good_total = done_total
        if self.world > 1 and torch.distributed.is_initialized():
            # a stage that stops while its neighbours wait in a hop would hang the job: everybody learns the worst status, whether
            # any stage failed in another way than its residency gate, and the FEWEST steps any stage completed
            flag = torch.tensor([status, status if status != 0x700 else 0, -done_total], device=m.device, dtype=torch.int64)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            worst, other, good_total = int(flag[0].item()), int(flag[1].item()), -int(flag[2].item())
            if other != 0:
                be.session_disable_engine()
                self._graphs, self._use_graph = {}, False
                raise RuntimeError(f"persistent decode engine: status 0x{other:x} on a pipeline stage; this generation is lost, "
                                   "the engine is now off for this process and later calls take the launch path")
            if worst == 0x700:
                # LOCK-STEP ROLLBACK.  A stage whose engine launch failed its residency gate wrote nothing from that step on
                # (status 0x700, include/mistral_hip.h) and kept forwarding its stale activations; the other stages ran ahead
                # on them.  Every stage knows from the reduction above how many steps completed EVERYWHERE: all stages rewind
                # to that step - positions, step counter, the input id of stage 0 - and run the rest again on the launch
                # path, hop by hop.  The steps that ran ahead only wrote ring slots (pos % W) of positions that are run again,
                # and those slots are rewritten with the right rows before anything reads them.
                logging.warning("persistent decode engine: residency gate failed on a pipeline stage after %d of %d steps; every "
                                "stage re-runs the rest on the launch path", good_total, issued_total)
                self._rollback(good_total)
                status, done_total = 0, issued_total
        if status != 0:
            missing = issued_total - done_total
            if status != 0x700:
                # a bounded wait timed out mid-step: the step's outputs are undefined and a raised status word makes every
                # later engine launch on this workspace leave at once.  Leave the process usable: clear the word, switch to
                # the launch path, drop the graphs that hold engine launches - the NEXT generate() runs launch by launch.
                be.session_disable_engine()
                self._graphs = {}
                self._use_graph = False
                raise RuntimeError(f"persistent decode engine: bounded wait 0x{status:x} timed out; this generation is "
                                   "lost (its cache is undefined), the engine is now off for this process and later calls "
                                   "take the launch path")
            logging.warning("persistent decode engine: %d of the GPU's workgroups were not resident together; %d step(s) "
                            "re-run on the launch path (engine off until the next generation probes the device again)",
                            st["arrivals"], missing)
            self._recover(missing)
        elif done_total != issued_total:
            raise RuntimeError(f"decode steps issued {issued_total} != completed {done_total}")
        first = self._collected()
        idx = torch.arange(first, first + n, device=m.device) + self._base
        idx = idx % self.HIST
        toks, lps = self.buf.hist_tok[idx], self.buf.hist_lp[idx]
        if self.world > 1:  # (the step counters of the stages run in lockstep, but only the last stage's rings hold samples)
            toks, lps = toks.contiguous(), lps.contiguous()
            m.pp_comm.broadcast(toks, src=self.world - 1)
            m.pp_comm.broadcast(lps, src=self.world - 1)
        if bool((toks == 0x7FFFFFFF).any()):
            # the argmax reductions start from "no index yet" and a row of NaN logits never replaces it (torch.argmax would
            # return the NaN's position): report what happened instead of the IndexError the next step's embedding raises
            raise FloatingPointError("greedy sample over a logits row without a maximum (NaN logits?)")
        self._pending -= n
        self._n_collected = first + n
        return toks, lps

    def _collected(self) -> int:
        return self._n_collected

    def _issued(self) -> int:
        return self._n_collected + self._pending

    def _rollback(self, good_total: int) -> None:
        """Pipeline stages, after a residency failure somewhere: rewind THIS stage to the state after `good_total` steps of the
        session (the number every stage completed) and run the remaining issued steps again on the launch path."""
        m, cache, be = self.model, self.cache, self.model._backend
        redo = self._issued() - good_total
        assert redo > 0 and self._base is not None
        be.session_disable_engine()                # clears the status word; this process takes the launch path from here on
        self._graphs = {}                          # (they hold engine launches)
        cache._seen = [p - redo for p in cache._seen]
        assert cache.kv_seqlens is not None
        cache.kv_seqlens.copy_(torch.tensor(cache._seen, dtype=torch.long))   # (stages that ran ahead advanced theirs)
        be.session_rewind(self._base + good_total)  # history-ring row / Philox offset of the next step
        # stage 0's next input id: the sample of step good_total - 1 (the last stage's history holds it), or the session's first
        tok = self._first.clone() if good_total == 0 else self.buf.hist_tok[(self._base + good_total - 1) % self.HIST].clone()
        if good_total > 0:
            m.pp_comm.broadcast(tok, src=self.world - 1)
        self.buf.tok.copy_(tok)
        for _ in range(redo):
            self._step_eager()
            cache.advance_host([1] * self.B)
        if m.device.type == "cuda":
            torch.cuda.synchronize(m.device)
        if be.session_status()["status"] != 0:
            raise RuntimeError("decode steps failed again on the launch path after a pipeline rollback")

    def _recover(self, missing: int) -> None:
        """The device state is that of the first failed step (nothing was written since): clear the status, switch
        this process to the launch path and run the `missing` steps again."""
        m, cache = self.model, self.cache
        m._backend.session_disable_engine()
        self._graphs = {}  # they hold engine launches
        cache._seen = [p - missing for p in cache._seen]
        for _ in range(missing):
            self._step_eager()
            cache.advance_host([1] * self.B)
        torch.cuda.synchronize(m.device)
