"""What `generate()` and the CLI need from a model (the role of the reference's model.py:11-43).

Four members are the reference's contract and stay abstract: `dtype`, `device`, `forward`, `from_folder`.  Two hooks
are this implementation's own and have working defaults, so a model that does not care simply inherits them:

  * `graphed_decode(cache)`  - context manager inside which single-token `forward` calls may be replayed from a
                               captured hipGraph (default: does nothing, calls run launch by launch);
  * `supports_prompt_logprobs` - whether `prompt_logprobs(...)` (log-probabilities of the prompt's next tokens without
                               the [T, vocab] logits tensor) exists; `generate()` falls back to `forward` otherwise.
"""
import contextlib
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Iterator, List, Optional, Union

import torch
from torch import nn


class ModelBase(nn.Module, ABC):
    supports_prompt_logprobs: bool = False

    # ---- the reference's contract -------------------------------------------------------------------------------
    @property
    @abstractmethod
    def dtype(self) -> torch.dtype:
        """Storage dtype of the weights (the HIP kernels take bf16 only)."""

    @property
    @abstractmethod
    def device(self) -> torch.device:
        """Device that owns the weights of this pipeline stage."""

    @abstractmethod
    def forward(self, input_ids: torch.Tensor, seqlens: List[int], cache=None) -> torch.Tensor:
        """Logits [sum(seqlens), vocab] for the concatenated sequences; `cache` is the caller's BufferCache."""

    @staticmethod
    @abstractmethod
    def from_folder(folder: Union[Path, str], max_batch_size: int = 1, num_pipeline_ranks: int = 1,
                    device: Union[torch.device, str] = "cuda", dtype: Optional[torch.dtype] = None) -> "ModelBase":
        """Build from `params.json` + a consolidated checkpoint in `folder` (this rank's tensors only)."""

    # ---- hooks with defaults --------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def graphed_decode(self, cache) -> Iterator["ModelBase"]:
        yield self
