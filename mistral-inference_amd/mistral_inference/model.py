"""ModelBase interface (reference model.py:11-43)."""
from abc import ABC, abstractmethod
from pathlib import Path
from typing import List, Optional, Union

import torch
import torch.nn as nn


class ModelBase(nn.Module, ABC):
    def __init__(self) -> None:
        super().__init__()

    @property
    @abstractmethod
    def dtype(self) -> torch.dtype: ...

    @property
    @abstractmethod
    def device(self) -> torch.device: ...

    @abstractmethod
    def forward(self, input_ids: torch.Tensor, seqlens: List[int], cache=None) -> torch.Tensor: ...

    @staticmethod
    @abstractmethod
    def from_folder(folder: Union[Path, str], max_batch_size: int = 1, num_pipeline_ranks: int = 1,
                    device: Union[torch.device, str] = "cuda", dtype: Optional[torch.dtype] = None) -> "ModelBase": ...
