"""CPU: the oracle's whole-sequence VAE restatement against goldens from the reference's chunked / cached WanVAE_."""
import os

import numpy as np
import pytest
import torch

from oracle import vae as ovae

CASES = ["f9_32x32", "f5_48x40", "f1_32x32", "f17_16x24"]


@pytest.fixture(scope="module")
def weights():
    return ovae.random_weights(seed=5)


@pytest.mark.parametrize("name", CASES)
def test_encode_decode_match_twin(name, weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_vae.npz"))
    mu = ovae.encode_mode(weights, torch.from_numpy(g[f"{name}_x"]))
    dec = ovae.decode(weights, torch.from_numpy(g[f"{name}_z"]))
    e_mu = np.abs(mu.numpy() - g[f"{name}_mu"]).max()
    e_dec = np.abs(dec.numpy() - g[f"{name}_dec"]).max()
    print(name, "max abs err: mu", e_mu, "dec", e_dec)
    assert mu.shape == g[f"{name}_mu"].shape and dec.shape == g[f"{name}_dec"].shape
    assert e_mu <= 2e-5 and e_dec <= 2e-5


def test_plans_and_shapes():
    sh = ovae.param_shapes()
    assert sum(int(np.prod(s)) for s in sh.values()) == 126_892_531  # the 127 M-parameter real config
    assert [k for k, *_ in ovae.decoder_plan()].count("up3d") == 2 and [k for k, *_ in ovae.encoder_plan()].count("down3d") == 2


@pytest.mark.parametrize("name", CASES)
def test_encode_decode_match_executed_diffusers_class(name, weights, golden_dir):
    """g8b: recorded from diffusers' AutoencoderKLWan (the class the reference executes, INFER:185-189; vendored copy
    longcat_video/modules/autoencoder_kl_wan.py) with the same weights loaded through the diffusers key map: chunked _encode
    (:1145-1170), decode with the clamp (:1222)."""
    g, b = np.load(os.path.join(golden_dir, "g8_vae.npz")), np.load(os.path.join(golden_dir, "g8b_vae_akw.npz"))
    mu = ovae.encode_mode(weights, torch.from_numpy(g[f"{name}_x"]))
    dec = ovae.decode(weights, torch.from_numpy(g[f"{name}_z"]))
    assert np.abs(mu.numpy() - b[f"{name}_mu"]).max() <= 2e-5 and np.abs(dec.numpy() - b[f"{name}_dec"]).max() <= 2e-5
    assert np.abs(b[f"{name}_dec"]).max() <= 1.0
