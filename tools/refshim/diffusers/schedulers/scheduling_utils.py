from dataclasses import dataclass
from enum import Enum

import torch


class KarrasDiffusionSchedulers(Enum):
    DDIMScheduler = 1
    UniPCMultistepScheduler = 13


class SchedulerMixin:
    pass


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
