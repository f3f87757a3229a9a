import torch.nn as nn


def get_activation(name):
    return {"silu": nn.SiLU, "swish": nn.SiLU, "gelu": nn.GELU, "relu": nn.ReLU, "mish": nn.Mish}[name.lower()]()
