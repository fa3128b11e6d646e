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
    hidden_size: int
    num_channels: int
    image_size: int
    patch_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    rope_theta: float = 1e4
    image_token_id: int = 10
    adapter_bias: bool = True
    spatial_merge_size: int = 1
    add_pre_mm_projector_layer_norm: bool = False
    mm_projector_id: str = ""


@dataclass
class TransformerArgs(_FromDict):
    dim: int
    n_layers: int
    head_dim: int
    hidden_dim: int
    n_heads: int
    n_kv_heads: int
    norm_eps: float
    vocab_size: int

    max_batch_size: int = 0
    rope_theta: Optional[float] = None          # None -> 1e6 (reference transformer.py:115)
    moe: Optional[MoeArgs] = None
    lora: Optional[LoraArgs] = None
    sliding_window: Union[None, int, List[Optional[int]]] = None
    _sliding_window: Union[None, int, List[Optional[int]]] = None
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
