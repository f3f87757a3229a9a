import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers semantics: a CPU generator draws on the CPU, the result is then moved to `device`."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    rand_device = device
    if generator is not None:
        gen_device = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_device != device.type and gen_device == "cpu":
            rand_device = torch.device("cpu")
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)
