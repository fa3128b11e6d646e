"""params.json schema (reference args.py:12-59, moe.py:10-13, lora.py:12-27), decoded without
simple_parsing: unknown keys are dropped with a warning like simple_parsing's from_dict does."""
from __future__ import annotations

import dataclasses
import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

PATCH_MERGE = "patch_merge"


def _decode(cls, d: Dict[str, Any]):
    names = {f.name for f in dataclasses.fields(cls)}
    kw = {}
    for k, v in d.items():
        if k not in names:
            logging.warning("%s: dropping unknown key %r", cls.__name__, k)
            continue
        sub = _NESTED.get((cls.__name__, k))
        kw[k] = _decode(sub, v) if (sub is not None and isinstance(v, dict)) else v
    return cls(**kw)


class _FromDict:
    @classmethod
    def from_dict(cls, d: Dict[str, Any], drop_extra_fields: Optional[bool] = None):
        return _decode(cls, dict(d))

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)


@dataclass
class MoeArgs(_FromDict):
    num_experts: int
    num_experts_per_tok: int


@dataclass
class LoraArgs(_FromDict):
    rank: int
    scaling: float

    def __post_init__(self) -> None:
        assert self.rank > 0
        assert self.scaling > 0.0


@dataclass
class VisionEncoderArgs(_FromDict):
    """`vision_encoder` block of params.json (Pixtral-12B, Mistral-Small-3.1)."""
    hidden_size: int            # tower width; heads are hidden_size / num_attention_heads wide (64 in shipped models)
    num_channels: int           # image channels (3)
    image_size: int             # largest side in pixels -> image_size / patch_size rotary positions per axis
    patch_size: int             # square patch = stride of the patch convolution (16 or 14)
    intermediate_size: int      # SwiGLU width of the tower's blocks
    num_hidden_layers: int
    num_attention_heads: int
    rope_theta: float = 1e4     # base of the 2-D rotary table
    image_token_id: int = 10    # placeholder id whose embedding rows are replaced by image features
    adapter_bias: bool = True   # vision_language_adapter Linear layers carry a bias
    spatial_merge_size: int = 1             # s: the patch merger folds s x s patches into one token
    add_pre_mm_projector_layer_norm: bool = False  # RMSNorm between tower and merger/adapter
    mm_projector_id: str = ""               # "patch_merge" enables the merger


@dataclass
class TransformerArgs(_FromDict):
    """Top level of params.json.  The first eight fields are mandatory; the kernels additionally require head_dim == 128,
    n_heads % n_kv_heads == 0 and dim, hidden_dim multiples of 8 (checked by libmistral_hip before any launch)."""
    dim: int                    # residual stream width D
    n_layers: int               # global layer count (a pipeline rank builds only its own contiguous range)
    head_dim: int
    hidden_dim: int             # SwiGLU width F (per expert for MoE)
    n_heads: int                # query heads H
    n_kv_heads: int             # key/value heads Hkv (GQA ratio H / Hkv)
    norm_eps: float
    vocab_size: int

    max_batch_size: int = 0     # set by from_folder; forward asserts len(seqlens) <= max_batch_size
    rope_theta: Optional[float] = None          # None -> 1e6 (reference transformer.py:115)
    moe: Optional[MoeArgs] = None               # sparse FFN: experts and experts per token
    lora: Optional[LoraArgs] = None             # un-merged LoRA layers are rejected; load_lora() merges adapters
    sliding_window: Union[None, int, List[Optional[int]]] = None   # one window, or one per layer (cycled)
    _sliding_window: Union[None, int, List[Optional[int]]] = None  # legacy spelling of the same key
    model_type: str = "transformer"
    vision_encoder: Optional[VisionEncoderArgs] = None

    def __post_init__(self) -> None:
        assert self.model_type == "transformer", self.model_type
        assert self.sliding_window is None or self._sliding_window is None
        # same aliasing as the reference (args.py:55-59)
        self.sliding_window = self.sliding_window if self.sliding_window is not None else self._sliding_window


_NESTED = {
    ("TransformerArgs", "moe"): MoeArgs,
    ("TransformerArgs", "lora"): LoraArgs,
    ("TransformerArgs", "vision_encoder"): VisionEncoderArgs,
}
