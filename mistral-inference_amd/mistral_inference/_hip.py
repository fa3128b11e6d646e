"""ctypes binding of libmistral_hip.so (C ABI: include/mistral_hip.h).

There is deliberately no fallback: if the library is missing or a tensor is not a contiguous bf16
device tensor, the call raises.  torch is used only to own device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
def _default_lib() -> str:
    """In-tree build (../lib, what build_native.py writes) first, then a copy shipped inside the package (wheel layout)."""
    in_tree = os.path.join(os.path.dirname(_HERE), "lib", "libmistral_hip.so")
    packaged = os.path.join(_HERE, "libmistral_hip.so")
    return packaged if (not os.path.exists(in_tree) and os.path.exists(packaged)) else in_tree


LIB_PATH = os.environ.get("MISTRAL_HIP_LIB", _default_lib())

MI_ABI_VERSION = 7
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU, EPI_LOGITS = 0, 1, 2, 3
BRANCH_NOCACHE, BRANCH_PREFILL, BRANCH_DECODE = 0, 1, 2
GEMV_MAX_T = 8
MI_ERR_SHAPE = -2
# storage dtypes of mi_forward_generic (include/mistral_hip.h `enum mi_dtype`)
DTYPE_CODES = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}

_vp = C.c_void_p


class MiLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("attention_norm", "wq", "wk", "wv", "wo", "ffn_norm", "w1", "w2", "w3", "gate",
                                   "expert_w_dev", "expert_w_host")]


class MiModel(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32),
        ("hidden_dim", C.c_int32), ("vocab_size", C.c_int32), ("n_layers", C.c_int32),
        ("num_experts", C.c_int32), ("top_k", C.c_int32), ("norm_eps", C.c_float),
        ("tok_embeddings", _vp), ("final_norm", _vp), ("output", _vp), ("rope_cs", _vp), ("rope_len", C.c_int32),
        ("layers", C.POINTER(MiLayer)),
    ]


class MiBatch(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("B", C.c_int32), ("branch", C.c_int32), ("max_q_len", C.c_int32),
        ("input_ids", _vp), ("q_start", _vp), ("kv_before", _vp), ("tok_seq", _vp), ("tok_pos", _vp),
        ("kv_seqlens", _vp), ("cache_k", C.POINTER(_vp)), ("cache_v", C.POINTER(_vp)),
        ("cache_sizes", C.POINTER(C.c_int32)), ("h", _vp), ("logits", _vp), ("workspace", _vp),
        ("workspace_bytes", C.c_size_t),
        ("greedy_token", _vp), ("greedy_logprob", _vp), ("hist_token", _vp), ("hist_logprob", _vp), ("hist_len", C.c_int32),
        ("sample_temperature", C.c_float), ("sample_top_p", C.c_float), ("sample_seed", C.c_uint64), ("sample_offset", C.c_uint64),  # ABI v5
        ("kv_layout", C.c_int32),  # ABI v7: KV_SLOT_MAJOR / KV_HEAD_MAJOR, the layout of every ring in cache_k / cache_v
    ]


_lib: Optional[C.CDLL] = None

_SIGS = {
    "mi_abi_version": (C.c_int, []),
    "mi_error_string": (C.c_char_p, [C.c_int]),
    "mi_last_error_detail": (C.c_char_p, []),
    "mi_embedding": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "mi_rmsnorm": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "mi_rope_inplace": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "mi_kv_write": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "mi_linear": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int), C.c_int,
                            _vp, _vp, C.c_float, _vp]),
    "mi_attn_decode_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_attn_decode": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "mi_attn_prefill": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp,
                                  _vp, C.c_int, C.c_float, C.c_int, _vp]),
    "mi_gelu": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "mi_greedy_sample": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "mi_sample_top_p": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp]),
    "mi_lm_head_logprobs_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mi_lm_head_logprobs": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "mi_moe_router": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_float, _vp]),
    "mi_qkv_rope_kvwrite": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp,
                                      C.c_float, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "mi_moe_experts_decode": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_float,
                                        _vp, _vp]),
    "mi_moe_grouped_gemm_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_moe_grouped_gemm": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                      C.c_size_t, _vp]),
    "mi_workspace_bytes": (C.c_size_t, [C.POINTER(MiModel), C.c_int, C.c_int, C.c_int]),
    "mi_forward": (C.c_int, [C.POINTER(MiModel), C.POINTER(MiBatch), _vp]),
    "mi_workspace_bytes_generic": (C.c_size_t, [C.POINTER(MiModel), C.c_int, C.c_int]),  # ABI v6
    "mi_forward_generic": (C.c_int, [C.POINTER(MiModel), C.POINTER(MiBatch), C.c_int, _vp]),
    "mi_embedding_generic": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "mi_rmsnorm_generic": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "mi_linear_generic": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int), C.c_int, _vp,
                                    C.c_int, _vp]),
    "mi_rope_inplace_generic": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "mi_attention_nocache_generic": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "mi_swiglu_generic": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "mi_gelu_generic": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "mi_set_decode_engine": (C.c_int, [C.c_int]),
    "mi_decode_engine_census": (C.c_int, [C.c_int]),
    "mi_decode_engine_reset": (C.c_int, [_vp, _vp]),
    "mi_decode_engine_status": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "mi_rccl_unique_id": (C.c_int, [_vp]),
    "mi_rccl_init": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _vp]),
    "mi_rccl_destroy": (C.c_int, [_vp]),
    "mi_rccl_send": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "mi_rccl_recv": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "mi_rccl_bcast": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "mi_rccl_group_start": (C.c_int, []),
    "mi_rccl_group_end": (C.c_int, []),
    "mi_rccl_last_error": (C.c_char_p, []),
    "mi_debug_engine_trace_bytes": (C.c_size_t, []),
    "mi_debug_set_engine_trace": (C.c_int, [_vp]),
    "mi_debug_set_engine_knobs": (C.c_int, [C.c_int, C.c_int]),
    "mi_debug_engine_sabotage": (C.c_int, [_vp, C.c_int, _vp]),
    "mi_debug_set_engine_holders": (C.c_int, [C.c_int]),
    "mi_debug_set_engine_variant": (C.c_int, [C.c_int]),
    "mi_debug_set_prefill_kernels": (C.c_int, [C.c_int, C.c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Load the library once.  Raises (never falls back) when it is absent or of the wrong ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python mistral-inference_amd/build_native.py` "
                "(hipcc --offload-arch=gfx950).  mistral_inference has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if handle.mi_abi_version() != MI_ABI_VERSION:
            raise RuntimeError(f"libmistral_hip ABI {handle.mi_abi_version()} != expected {MI_ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        L = lib()
        detail = L.mi_last_error_detail().decode()
        if rc == -5 or "rccl" in what:  # mi_rccl_* keep their own text (a failed dlopen of librccl returns MI_ERR_UNSUPPORTED)
            detail = (L.mi_rccl_last_error().decode() or detail)
        raise RuntimeError(f"libmistral_hip {what}: {L.mi_error_string(rc).decode()} [{detail}] (code {rc})")


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def dev_ptr(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = torch.bfloat16) -> Optional[int]:
    """Raw device pointer of a tensor the kernels may read as dense rows."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mistral_inference runs only on a HIP device; got a CPU tensor "
                           "(there is no CPU fallback -- use the reference or oracle/ for CPU runs)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype} (the gfx950 kernels are bf16-storage only)")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise RuntimeError("innermost dimension must be contiguous")
    return t.data_ptr()


# ------------------------------------------------------------------------------------------------
# leaf operator wrappers (torch tensors in, torch tensors out; all work done by the library)
# ------------------------------------------------------------------------------------------------
def _generic_code(dt: torch.dtype) -> int:
    if dt not in DTYPE_CODES:
        raise RuntimeError(f"storage dtype {dt}: the HIP kernels take bfloat16, float16 and float32")
    return DTYPE_CODES[dt]


def check_ids_on_host(ids: torch.Tensor, vocab: int) -> None:
    """nn.Embedding's IndexError for ids the host can see without a device synchronisation."""
    if ids.device.type == "cpu" and ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= vocab):
        raise IndexError("index out of range in self")


def embedding(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    T, (V, D) = ids.numel(), table.shape
    check_ids_on_host(ids, V)
    out = torch.empty((T, D), dtype=table.dtype, device=table.device)
    ids = ids.to(device=table.device, dtype=torch.long).contiguous()
    dt = table.dtype
    if dt != torch.bfloat16:  # fp16 / fp32 storage: the generic kernels (csrc/generic.hip)
        check(lib().mi_embedding_generic(dev_ptr(out, dt), dev_ptr(table, dt), dev_ptr(ids, torch.long), T, D, V, _generic_code(dt),
                                         stream_ptr(table.device)), "mi_embedding_generic")
        return out
    check(lib().mi_embedding(dev_ptr(out), dev_ptr(table), dev_ptr(ids, torch.long), T, D, V, stream_ptr(table.device)),
          "mi_embedding")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    o = torch.empty_like(x2) if out is None else out
    dt = x.dtype
    if dt != torch.bfloat16:
        check(lib().mi_rmsnorm_generic(dev_ptr(o, dt), dev_ptr(x2, dt), dev_ptr(w, dt), x2.shape[0], x2.shape[1], float(eps),
                                       _generic_code(dt), stream_ptr(x.device)), "mi_rmsnorm_generic")
        return o.view(x.shape)
    check(lib().mi_rmsnorm(dev_ptr(o), dev_ptr(x2), dev_ptr(w), x2.shape[0], x2.shape[1], float(eps),
                           stream_ptr(x.device)), "mi_rmsnorm")
    return o.view(x.shape)


def linear(x: torch.Tensor, weights: Sequence[torch.Tensor], epilogue: int = EPI_STORE,
           residual: Optional[torch.Tensor] = None, norm_w: Optional[torch.Tensor] = None, eps: float = 0.0,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(x @ cat(weights)^T); SWIGLU: weights = (W1, W3)."""
    assert 1 <= len(weights) <= 3 and x.dim() == 2
    M, K = x.shape
    n_rows = [w.shape[0] for w in weights]
    for w in weights:
        assert w.shape[1] == K and w.is_contiguous()
    N = n_rows[0] if epilogue == EPI_SWIGLU else sum(n_rows)
    if x.dtype != torch.bfloat16:
        return _linear_generic(x, weights, n_rows, N, epilogue, residual, norm_w, eps, out)
    odt = torch.float32 if epilogue == EPI_LOGITS else torch.bfloat16
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=x.device)
    wp = (_vp * 3)(*[dev_ptr(w) for w in weights], *([None] * (3 - len(weights))))
    nr = (C.c_int * 3)(*n_rows, *([0] * (3 - len(weights))))
    check(lib().mi_linear(dev_ptr(out, odt), out.stride(0), dev_ptr(x), x.stride(0), M, K, wp, nr, epilogue,
                          dev_ptr(residual), dev_ptr(norm_w), float(eps), stream_ptr(x.device)), "mi_linear")
    return out


def _linear_generic(x, weights, n_rows, N, epilogue, residual, norm_w, eps, out):
    """`linear` for fp16 / fp32 storage (csrc/generic.hip): the same epilogues composed from mi_*_generic calls."""
    dt, dev = x.dtype, x.device
    code, st = _generic_code(dt), stream_ptr(dev)
    M, K = x.shape
    L = lib()
    if norm_w is not None:
        x = rmsnorm(x, norm_w, eps)
    if not x.is_contiguous():
        x = x.contiguous()

    def run(ws, rows, epi, o, res=None):
        wp = (_vp * 3)(*[dev_ptr(w, dt) for w in ws], *([None] * (3 - len(ws))))
        nr = (C.c_int * 3)(*rows, *([0] * (3 - len(ws))))
        odt = torch.float32 if epi == EPI_LOGITS else dt
        check(L.mi_linear_generic(dev_ptr(o, odt), o.stride(0), dev_ptr(x, dt), x.stride(0), M, K, wp, nr, epi, dev_ptr(res, dt), code, st),
              "mi_linear_generic")
        return o
    if epilogue == EPI_SWIGLU:
        a = run(weights[:1], n_rows[:1], EPI_STORE, torch.empty((M, N), dtype=dt, device=dev) if out is None else out)
        b = run(weights[1:2], n_rows[1:2], EPI_STORE, torch.empty((M, N), dtype=dt, device=dev))
        assert a.is_contiguous()
        check(L.mi_swiglu_generic(dev_ptr(a, dt), dev_ptr(b, dt), M, N, code, st), "mi_swiglu_generic")
        return a
    odt = torch.float32 if epilogue == EPI_LOGITS else dt
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=dev)
    if epilogue == EPI_RESIDUAL:
        assert residual is not None and residual.stride(0) == out.stride(0), "residual rows must have the output's stride"
    return run(weights, n_rows, epilogue, out, residual if epilogue == EPI_RESIDUAL else None)


def rope_inplace(qkv: torch.Tensor, n_heads: int, n_kv_heads: int, head_dim: int, rope_cs: torch.Tensor,
                 tok_pos: torch.Tensor) -> None:
    assert rope_cs.dtype == torch.float32 and rope_cs.is_contiguous() and tok_pos.dtype == torch.int32
    if qkv.dtype != torch.bfloat16:
        check(lib().mi_rope_inplace_generic(dev_ptr(qkv, qkv.dtype), qkv.stride(0), qkv.shape[0], (n_heads + n_kv_heads) * head_dim,
                                            head_dim, dev_ptr(rope_cs, torch.float32), dev_ptr(tok_pos, torch.int32),
                                            _generic_code(qkv.dtype), stream_ptr(qkv.device)), "mi_rope_inplace_generic")
        return
    check(lib().mi_rope_inplace(dev_ptr(qkv), qkv.stride(0), qkv.shape[0], n_heads, n_kv_heads, head_dim,
                                dev_ptr(rope_cs, torch.float32), rope_cs.shape[0], dev_ptr(tok_pos, torch.int32),
                                stream_ptr(qkv.device)), "mi_rope_inplace")


KV_SLOT_MAJOR, KV_HEAD_MAJOR = 0, 1  # include/mistral_hip.h: MI_KV_SLOT_MAJOR / MI_KV_HEAD_MAJOR


def kv_layout_of(ring: torch.Tensor) -> int:
    """Layout of a K/V ring given in the reference's SHAPE [max_batch, W, n_kv_heads, head_dim] (cache.py:163-167): contiguous =
    the reference's layout; a permuted view of a contiguous [max_batch, n_kv_heads, W, head_dim] tensor = head-major (what
    `BufferCache` allocates: one kv head's slots are contiguous).  Anything else is rejected."""
    assert ring.ndim == 4, f"K/V ring: expected [max_batch, W, n_kv_heads, head_dim], got {tuple(ring.shape)}"
    B, W, H, D = ring.shape
    st = ring.stride()
    if ring.is_contiguous():
        return KV_SLOT_MAJOR  # (also every degenerate shape whose two layouts coincide)
    if st[3] == 1 and st[1] == D and st[2] == W * D and (B == 1 or st[0] == H * W * D):
        return KV_HEAD_MAJOR
    raise ValueError(f"K/V ring of shape {tuple(ring.shape)} with strides {st}: neither [B, W, H, D] contiguous nor a "
                     "permuted view of a contiguous [B, H, W, D] tensor")


def _kv_layout2(cache_k: torch.Tensor, cache_v: torch.Tensor) -> int:
    assert cache_k.shape == cache_v.shape and cache_k.stride() == cache_v.stride(), "cache_k and cache_v must share shape and layout"
    return kv_layout_of(cache_k)


def kv_write(cache_k: torch.Tensor, cache_v: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tok_seq: torch.Tensor,
             tok_pos: torch.Tensor, q_start: torch.Tensor) -> None:
    W = cache_k.shape[1]
    kv_dim = cache_k.shape[2] * cache_k.shape[3]
    assert k.stride(0) == v.stride(0)
    check(lib().mi_kv_write(dev_ptr(cache_k), dev_ptr(cache_v), W, dev_ptr(k), dev_ptr(v), k.stride(0), k.shape[0], kv_dim,
                            dev_ptr(tok_seq, torch.int32), dev_ptr(tok_pos, torch.int32), dev_ptr(q_start, torch.int32),
                            _kv_layout2(cache_k, cache_v), cache_k.shape[3], stream_ptr(k.device)), "mi_kv_write")


_decode_scratch = {}


def attn_decode(q: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, n_heads: int, tok_pos: torch.Tensor
                ) -> torch.Tensor:
    B = q.shape[0]
    _, W, Hkv, Dh = cache_k.shape
    need = lib().mi_attn_decode_scratch_bytes(B, n_heads, Hkv, Dh, W)
    key = (q.device, need)
    if key not in _decode_scratch:
        _decode_scratch[key] = torch.zeros(need, dtype=torch.uint8, device=q.device)
    out = torch.empty((B, n_heads * Dh), dtype=q.dtype, device=q.device)
    check(lib().mi_attn_decode(dev_ptr(out), dev_ptr(q), q.stride(0), dev_ptr(cache_k), dev_ptr(cache_v), W, B, n_heads, Hkv,
                               Dh, dev_ptr(tok_pos, torch.int32), dev_ptr(_decode_scratch[key], torch.uint8),
                               _kv_layout2(cache_k, cache_v), stream_ptr(q.device)), "mi_attn_decode")
    return out


def attn_prefill(qkv: torch.Tensor, n_heads: int, n_kv_heads: int, head_dim: int, cache_k: Optional[torch.Tensor],
                 cache_v: Optional[torch.Tensor], W: int, q_start: Optional[torch.Tensor],
                 kv_before: Optional[torch.Tensor], B: int, max_q_len: int, causal: bool = True,
                 softmax_scale: float = 0.0) -> torch.Tensor:
    """softmax_scale <= 0: head_dim ** -0.5."""
    T = qkv.shape[0]
    out = torch.empty((T, n_heads * head_dim), dtype=qkv.dtype, device=qkv.device)
    if qkv.dtype != torch.bfloat16:
        if causal or cache_k is not None:
            raise RuntimeError("fp16 / fp32 storage: the stand-alone attention operator exists for the cache=None form only "
                               "(cached attention runs inside mi_forward_generic)")
        check(lib().mi_attention_nocache_generic(dev_ptr(out, qkv.dtype), dev_ptr(qkv, qkv.dtype), qkv.stride(0), T, n_heads, n_kv_heads,
                                                 head_dim, float(softmax_scale), _generic_code(qkv.dtype), stream_ptr(qkv.device)),
              "mi_attention_nocache_generic")
        return out
    check(lib().mi_attn_prefill(dev_ptr(out), dev_ptr(qkv), qkv.stride(0), dev_ptr(cache_k), dev_ptr(cache_v), W, B,
                                max_q_len, n_heads, n_kv_heads, head_dim, dev_ptr(q_start, torch.int32),
                                dev_ptr(kv_before, torch.int32), 1 if causal else 0, float(softmax_scale),
                                _kv_layout2(cache_k, cache_v) if cache_k is not None else KV_SLOT_MAJOR,
                                stream_ptr(qkv.device)), "mi_attn_prefill")
    return out


def lm_head_logprobs(x: torch.Tensor, w: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """fp32 [M]: log_softmax(float(bf16(x @ w^T)))[m, target[m]] without materialising the [M, vocab] logits."""
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and w.is_contiguous()
    M, K = x.shape
    V = w.shape[0]
    tgt = target.to(device=x.device, dtype=torch.int32).contiguous()
    assert tgt.numel() == M
    need = lib().mi_lm_head_logprobs_scratch_bytes(M, V)
    scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
    out = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib().mi_lm_head_logprobs(dev_ptr(out, torch.float32), dev_ptr(x), x.stride(0), M, K, dev_ptr(w), V,
                                    dev_ptr(tgt, torch.int32), dev_ptr(scratch, torch.uint8), need, stream_ptr(x.device)),
          "mi_lm_head_logprobs")
    return out


def greedy_sample(logits: torch.Tensor):
    """(token int64 [B], logprob fp32 [B]): first argmax of each fp32 logits row and log_softmax at it."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B, V = logits.shape
    tok = torch.empty(B, dtype=torch.long, device=logits.device)
    lp = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(lib().mi_greedy_sample(dev_ptr(logits, torch.float32), logits.stride(0), B, V, dev_ptr(tok, torch.long),
                                 dev_ptr(lp, torch.float32), stream_ptr(logits.device)), "mi_greedy_sample")
    return tok, lp


def sample_top_p(logits: torch.Tensor, temperature: float, top_p: float, seed: int = 0, offset: int = 0,
                 uniforms: Optional[torch.Tensor] = None):
    """(token int64 [B], logprob fp32 [B]): one nucleus draw per fp32 logits row (reference generate.py:151-170) in ONE launch;
    logprob = log_softmax(row)[token] of the unscaled logits.  `uniforms` (fp32 [B] in [0, 1)) replaces the Philox variate."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B, V = logits.shape
    tok = torch.empty(B, dtype=torch.long, device=logits.device)
    lp = torch.empty(B, dtype=torch.float32, device=logits.device)
    if uniforms is not None:
        uniforms = uniforms.to(device=logits.device, dtype=torch.float32).contiguous()
        assert uniforms.numel() == B
    check(lib().mi_sample_top_p(dev_ptr(logits, torch.float32), logits.stride(0), B, V, float(temperature), float(top_p),
                                int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), dev_ptr(uniforms, torch.float32),
                                dev_ptr(tok, torch.long), dev_ptr(lp, torch.float32), stream_ptr(logits.device)), "mi_sample_top_p")
    return tok, lp


def gelu_(x: torch.Tensor) -> torch.Tensor:
    """In-place exact GELU on a 2-D bf16 tensor."""
    assert x.dim() == 2
    if x.dtype != torch.bfloat16:
        check(lib().mi_gelu_generic(dev_ptr(x, x.dtype), x.stride(0), x.shape[0], x.shape[1], _generic_code(x.dtype),
                                    stream_ptr(x.device)), "mi_gelu_generic")
        return x
    check(lib().mi_gelu(dev_ptr(x), x.stride(0), x.shape[0], x.shape[1], stream_ptr(x.device)), "mi_gelu")
    return x


def moe_router(x: torch.Tensor, gate: torch.Tensor, top_k: int, norm_w: Optional[torch.Tensor] = None, eps: float = 0.0):
    T, D = x.shape
    E = gate.shape[0]
    idx = torch.empty((T, top_k), dtype=torch.int32, device=x.device)
    w = torch.empty((T, top_k), dtype=torch.float32, device=x.device)
    check(lib().mi_moe_router(dev_ptr(idx, torch.int32), dev_ptr(w, torch.float32), dev_ptr(x), x.stride(0), T, D,
                              dev_ptr(gate), E, top_k, dev_ptr(norm_w), float(eps), stream_ptr(x.device)), "mi_moe_router")
    return idx, w


def qkv_rope_kvwrite(x: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, head_dim: int, rope_cs: torch.Tensor,
                     tok_pos: torch.Tensor, norm_w: Optional[torch.Tensor] = None, eps: float = 0.0,
                     cache_k: Optional[torch.Tensor] = None, cache_v: Optional[torch.Tensor] = None,
                     tok_seq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode-sized (T <= 8) fused [RMSNorm] + q|k|v projection + RoPE [+ ring write at tok_pos % W of row tok_seq]:
    returns qkv [T, (H + 2 Hkv) * head_dim] (post-RoPE)."""
    T, D = x.shape
    nq, nkv = wq.shape[0], wk.shape[0]
    assert wv.shape[0] == nkv and rope_cs.dtype == torch.float32 and rope_cs.is_contiguous() and tok_pos.dtype == torch.int32
    out = torch.empty((T, nq + 2 * nkv), dtype=x.dtype, device=x.device)
    W = cache_k.shape[1] if cache_k is not None else 0
    check(lib().mi_qkv_rope_kvwrite(dev_ptr(out), out.stride(0), dev_ptr(x), x.stride(0), T, D, dev_ptr(wq), dev_ptr(wk),
                                    dev_ptr(wv), nq // head_dim, nkv // head_dim, head_dim, dev_ptr(norm_w), float(eps),
                                    dev_ptr(rope_cs, torch.float32), rope_cs.shape[0], dev_ptr(tok_pos, torch.int32),
                                    dev_ptr(tok_seq, torch.int32), dev_ptr(cache_k), dev_ptr(cache_v), W,
                                    _kv_layout2(cache_k, cache_v) if cache_k is not None else KV_SLOT_MAJOR,
                                    stream_ptr(x.device)), "mi_qkv_rope_kvwrite")
    return out


def moe_experts(x: torch.Tensor, expert_tab: torch.Tensor, n_experts: int, hidden_dim: int, sel_idx: torch.Tensor,
                sel_w: torch.Tensor, residual: Optional[torch.Tensor] = None, norm_w: Optional[torch.Tensor] = None,
                eps: float = 0.0) -> torch.Tensor:
    """bf16(residual + sum over each token's picked experts (ascending id) of bf16(w * expert(x))) (reference moe.py:28-32 +
    transformer_layers.py:168).  expert_tab: int64 device tensor [E, 3] of (w1, w2, w3) pointers; sel_idx/sel_w: the
    router's output.  T <= 8 runs the two weight-streaming launches (mi_moe_experts_decode), larger T the token-grouped
    MFMA GEMMs (mi_moe_grouped_gemm).  residual None: zeros (the bare MoeLayer.forward value)."""
    T, D = x.shape
    k = sel_idx.shape[1]
    if residual is None:
        residual = torch.zeros_like(x)
    out = torch.empty_like(x)
    L, st = lib(), stream_ptr(x.device)
    if T <= GEMV_MAX_T:
        hid = torch.empty((T * k, hidden_dim), dtype=x.dtype, device=x.device)
        check(L.mi_moe_experts_decode(dev_ptr(out), dev_ptr(residual), dev_ptr(x), x.stride(0), T, D, hidden_dim,
                                      expert_tab.data_ptr(), dev_ptr(sel_idx, torch.int32), dev_ptr(sel_w, torch.float32), k,
                                      dev_ptr(norm_w), float(eps), dev_ptr(hid), st), "mi_moe_experts_decode")
        return out
    assert norm_w is None, "the grouped path takes already-normalised rows"
    xc = x if x.is_contiguous() else x.contiguous()
    need = L.mi_moe_grouped_gemm_scratch_bytes(T, D, hidden_dim, n_experts, k)
    scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
    check(L.mi_moe_grouped_gemm(dev_ptr(out), dev_ptr(residual), dev_ptr(xc), D, T, D, hidden_dim, n_experts, k,
                                expert_tab.data_ptr(), dev_ptr(sel_idx, torch.int32), dev_ptr(sel_w, torch.float32),
                                scratch.data_ptr(), need, st), "mi_moe_grouped_gemm")
    return out


def set_decode_engine(enabled: bool) -> bool:
    """Persistent decode engine on/off (default on; see include/mistral_hip.h).  Returns the previous setting."""
    return bool(lib().mi_set_decode_engine(1 if enabled else 0))


def decode_engine_status(workspace: torch.Tensor) -> dict:
    """Control words of the persistent decode engine in `workspace` (synchronises the current stream)."""
    st = (C.c_uint32 * 8)()
    check(lib().mi_decode_engine_status(workspace.data_ptr(), stream_ptr(workspace.device), st), "mi_decode_engine_status")
    return {"epoch": int(st[0]), "status": int(st[1]), "abort": int(st[2]), "bad_id": int(st[3]), "engine_launches": int(st[4]),
            "steps": int(st[5]), "arrivals": int(st[6])}


def debug_engine_sabotage(workspace: torch.Tensor, launches: int) -> None:
    """Test hook: the next `launches` engine launches on this workspace fail their residency gate."""
    check(lib().mi_debug_engine_sabotage(workspace.data_ptr(), launches, stream_ptr(workspace.device)), "mi_debug_engine_sabotage")


def debug_set_prefill_kernels(attn_waves: int = -1, gemm_tail: int = -1) -> None:
    """Same-process A/B of the prefill kernels' alternative forms (include/mistral_hip_debug.h; negative = keep)."""
    check(lib().mi_debug_set_prefill_kernels(attn_waves, gemm_tail), "mi_debug_set_prefill_kernels")


def decode_engine_reset(workspace: torch.Tensor) -> None:
    """Clear a raised engine status (see include/mistral_hip.h: residency)."""
    check(lib().mi_decode_engine_reset(workspace.data_ptr(), stream_ptr(workspace.device)), "mi_decode_engine_reset")


def ptr_array(ptrs: List[Optional[int]]):
    return (_vp * len(ptrs))(*ptrs)
