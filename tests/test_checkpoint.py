"""CPU: checkpoint plumbing -- the safetensors reader (own parser; the `safetensors` package is only the writer here) and the
diffusers -> twin parameter-name maps of the VAE and the DiT (INFER:179-197 loads diffusers-layout checkpoints)."""
import json
import os

import numpy as np
import pytest
import torch

from worldforge_amd import checkpoint


def _tensors():
    g = torch.Generator().manual_seed(0)
    return {"a.weight": torch.randn(5, 7, generator=g), "a.bias": torch.randn(7, generator=g).to(torch.bfloat16),
            "b.idx": torch.arange(12, dtype=torch.int64).reshape(3, 4), "c.h": torch.randn(2, 3, 4, generator=g).to(torch.float16),
            "d.scalar": torch.tensor(3.5), "e.empty": torch.zeros(0, 4)}


def test_single_file_round_trip(tmp_path):
    from safetensors.torch import save_file
    t = _tensors()
    save_file(t, str(tmp_path / "diffusion_pytorch_model.safetensors"), metadata={"format": "pt"})
    got = checkpoint.load_dir(str(tmp_path))
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and torch.equal(got[k], t[k]), k
    sub = checkpoint.load_file(str(tmp_path / "diffusion_pytorch_model.safetensors"), ["a.bias"])
    assert list(sub) == ["a.bias"]
    with pytest.raises(KeyError):
        checkpoint.load_file(str(tmp_path / "diffusion_pytorch_model.safetensors"), ["nope"])


def test_sharded_index_round_trip(tmp_path):
    from safetensors.torch import save_file
    t = _tensors()
    names = sorted(t)
    shards = {"m-00001-of-00002.safetensors": names[:3], "m-00002-of-00002.safetensors": names[3:]}
    wm = {}
    for fn, ns in shards.items():
        save_file({n: t[n] for n in ns}, str(tmp_path / fn))
        wm.update({n: fn for n in ns})
    (tmp_path / "m.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))
    got = checkpoint.load_dir(str(tmp_path))
    assert set(got) == set(t) and all(torch.equal(got[k], t[k]) for k in t)
    os.remove(tmp_path / "m-00002-of-00002.safetensors")
    with pytest.raises(FileNotFoundError):
        checkpoint.load_dir(str(tmp_path))


def test_rejects_corrupt_files(tmp_path):
    p = tmp_path / "x.safetensors"
    p.write_bytes(b"\x00\x01")
    with pytest.raises(ValueError):
        checkpoint.load_file(str(p))
    p.write_bytes((10 ** 9).to_bytes(8, "little") + b"{}")
    with pytest.raises(ValueError):
        checkpoint.load_file(str(p))
    hdr = json.dumps({"w": {"dtype": "F32", "shape": [4], "data_offsets": [0, 8]}}).encode()
    p.write_bytes(len(hdr).to_bytes(8, "little") + hdr + b"\x00" * 8)
    with pytest.raises(ValueError):
        checkpoint.load_file(str(p))
    os.remove(p)
    with pytest.raises(FileNotFoundError):
        checkpoint.load_dir(str(tmp_path))  # no *.safetensors left


def test_vae_diffusers_key_map_covers_the_executed_class(golden_dir):
    """Every parameter of diffusers' AutoencoderKLWan (names + shapes recorded from the vendored class, g8b) maps to exactly one twin
    parameter of the same element count, and every twin parameter is hit."""
    from oracle import vae as ovae
    from worldforge_amd.vae import diffusers_key_map, diffusers_to_twin_state_dict
    b = np.load(os.path.join(golden_dir, "g8b_vae_akw.npz"))
    names = [str(n) for n in b["param_names"]]
    shapes = [tuple(int(v) for v in str(s).split(",")) if str(s) else () for s in b["param_shapes"]]
    fake = {n: torch.empty(s) for n, s in zip(names, shapes)}
    twin = diffusers_to_twin_state_dict(fake)
    want = ovae.param_shapes()
    assert set(twin) == set(want)
    assert all(twin[k].numel() == int(np.prod(want[k])) for k in want)
    assert len(set(diffusers_key_map().values())) == len(diffusers_key_map())  # injective
    with pytest.raises(KeyError):
        diffusers_to_twin_state_dict({"encoder.nope.weight": torch.empty(1)})


def test_dit_diffusers_key_map_hits_every_twin_parameter():
    """dit.diffusers_key_map values + the scale_shift_table renames cover exactly the twin's parameter set (oracle.dit.random_weights
    has the twin WanModel's state_dict keys: tools/make_goldens.py g_dit loads it with strict=True)."""
    from oracle import dit as odit
    from worldforge_amd.dit import diffusers_key_map
    cfg = odit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64)
    twin = set(odit.random_weights(cfg, seed=0))
    km = diffusers_key_map(cfg.num_layers)
    mapped = set()
    for d, t in km.items():
        for leaf in ("weight", "bias"):
            if f"{t}.{leaf}" in twin:
                mapped.add(f"{t}.{leaf}")
    mapped |= {"head.modulation"} | {f"blocks.{i}.modulation" for i in range(cfg.num_layers)}
    assert mapped == twin, sorted(twin ^ mapped)[:8]
    assert len(set(km.values())) == len(km)
