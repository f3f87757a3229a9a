import torch.nn as nn


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        """diffusers.ModelMixin.dtype: the dtype of the module's parameters (read at PIPE:733, SCHED:1284)."""
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device
