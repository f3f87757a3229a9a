from dataclasses import dataclass

import torch


@dataclass
class WanPipelineOutput:
    frames: torch.Tensor
