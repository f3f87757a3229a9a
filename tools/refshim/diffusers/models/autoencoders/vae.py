"""diffusers.models.autoencoders.vae stand-ins: the two small containers the Wan / LongCat VAE modules return."""
from dataclasses import dataclass

import torch

from ...utils import BaseOutput


@dataclass
class DecoderOutput(BaseOutput):
    sample: torch.Tensor = None


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
