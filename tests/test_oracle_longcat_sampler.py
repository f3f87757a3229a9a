"""CPU: oracle/longcat_sampler.py against trajectories recorded from the unmodified reference LongCatVideoPipeline.generate_i2v +
FlowMatchEulerDiscreteScheduler (tests/golden/g12_longcat_pipe_*.npz; tools/make_goldens.py longcat_pipe) with the deterministic
DiT / VAE stand-ins of tests/fakes.py."""
import os

import numpy as np
import pytest
import torch

from oracle import longcat_sampler as ols
from tests.fakes import FakeLongCatDiT, FakeVAE, synthetic_ref_and_mask

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "irr_flf": dict(steps=9, R=2, guide=8, rnd=8, flf=True, omega=1.8, omega_r=1.0, cfg=4.0, F=9, H=32, W=32, guided=True, shift=1.0,
                    distill=False, maxrep=None),
    "dsg_shift": dict(steps=6, R=3, guide=3, rnd=5, flf=False, omega=2.5, omega_r=1.5, cfg=3.0, F=5, H=32, W=48, guided=True, shift=8.0,
                      distill=False, maxrep=None),
    "plain": dict(steps=5, R=3, guide=0, rnd=0, flf=False, omega=1.8, omega_r=1.0, cfg=4.0, F=5, H=32, W=32, guided=False, shift=3.0,
                  distill=False, maxrep=None),
    "nocfg_distill": dict(steps=6, R=2, guide=6, rnd=6, flf=True, omega=1.8, omega_r=1.0, cfg=1.0, F=9, H=32, W=32, guided=True,
                          shift=1.0, distill=True, maxrep=2),
}


def case_inputs(c):
    g = torch.Generator().manual_seed(17)
    image = torch.rand(3, c["H"], c["W"], generator=g)
    ref, mask = synthetic_ref_and_mask(c["F"], c["H"], c["W"], seed=6)
    pe = torch.randn(1, 1, 12, 32, generator=g).to(torch.bfloat16)
    ne = torch.randn(1, 1, 12, 32, generator=g).to(torch.bfloat16)
    pm = torch.zeros(1, 12, dtype=torch.int64)
    pm[:, :9] = 1
    nm = torch.zeros(1, 12, dtype=torch.int64)
    nm[:, :4] = 1
    return image, ref, mask, pe, pm, ne, nm


def sampler_config(c, vae_dtype=torch.float32):
    return ols.LongCatSamplerConfig(vae_dtype=vae_dtype, num_inference_steps=c["steps"], guidance_scale=c["cfg"], shift=c["shift"], use_distill=c["distill"],
                                    guided=c["guided"], resample_steps=c["R"], guide_steps=c["guide"], resample_round=c["rnd"],
                                    omega=c["omega"], omega_resample=c["omega_r"], use_pca_channel_selection=c["flf"],
                                    max_replace_threshold=c["maxrep"])


# (case, VAE dtype): g12 = fp32 VAE stand-in; g12b = the stand-in in bf16, the dtype the LongCat entry loads its VAE in
# (run_longcat_worldforge_single.py:205) -- pins the `.to(vae.dtype)` hand-offs of fuse_latents and of the final decode
TRAJ = [(n, torch.float32) for n in CASES] + [("irr_flf", torch.bfloat16), ("nocfg_distill", torch.bfloat16)]


def golden_pipe(name, vae_dtype):
    f = f"g12_longcat_pipe_{name}.npz" if vae_dtype == torch.float32 else f"g12b_longcat_pipe_{name}_vaebf16.npz"
    return np.load(os.path.join(GOLD, f))


@pytest.mark.parametrize("name,vae_dtype", TRAJ)
def test_sampler_matches_reference_trajectory(name, vae_dtype):
    c = CASES[name]
    G = golden_pipe(name, vae_dtype)
    image, ref, mask, pe, pm, ne, nm = case_inputs(c)
    dit, vae = FakeLongCatDiT(), FakeVAE(vae_dtype)
    mean, std = vae.config.latents_mean, vae.config.latents_std
    gen = torch.manual_seed(42)
    sig, ts = ols.make_schedule(ols.timesteps_sigmas(c["steps"], c["distill"]), c["shift"])
    assert np.array_equal(sig.numpy(), G["sigmas"]) and np.array_equal(ts.numpy(), G["timesteps"])
    lat = ols.prepare_latents((2.0 * image - 1.0)[None], c["F"], lambda x, g: vae.encode(x).latent_dist.sample(g), mean, std, gen)
    assert np.array_equal(lat.numpy(), G["latents0"])
    do_cfg = c["cfg"] > 1.0
    trace = []
    out = ols.run(sampler_config(c, vae_dtype), latents=lat, dit=dit, prompt_embeds=torch.cat([ne, pe]) if do_cfg else pe,
                  prompt_mask=torch.cat([nm, pm]) if do_cfg else pm, video_ref=ref, mask=mask,
                  decode=lambda z: vae.decode(z)[0], encode_mode=lambda v: vae.encode(v).latent_dist.mode(), mean=mean, std=std,
                  generator=gen, trace=trace)
    steps = [t for t in trace if t[0] == "step"]
    # golden record order: per step i its resample rounds, then one more scheduler.step for the DSG update where one is made
    j, k = 0, 0
    for i in range(c["steps"]):
        rounds = c["R"] if (c["guided"] and i < c["rnd"]) else 1
        for r in range(rounds):
            _, si, sr, prev, x0 = steps[k]
            assert (si, sr) == (i, r)
            np.testing.assert_allclose(prev.numpy(), G[f"call{j}_prev"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(x0.numpy(), G[f"call{j}_x0"], rtol=0, atol=2e-5)
            j, k = j + 1, k + 1
        if c["guided"] and i < c["rnd"] and rounds > 1:
            j += 1
    assert j == int(G["n_step_calls"][0]) and k == len(steps)
    assert [dit.calls, vae.n_enc, vae.n_dec] == (G["n_calls"] - np.array([0, 0, 1])).tolist()  # the final decode comes next
    frames = ols.decode_final(out, lambda z: vae.decode(z)[0], mean, std, vae_dtype)
    np.testing.assert_allclose(frames.numpy(), G["frames"], rtol=0, atol=2e-5)


def test_longcat_selection_rule():
    sims = [0.9, 0.2, 0.8, 0.1, 0.85, 0.88, 0.3, 0.86]
    assert ols.select_from_similarities(sims, 1) == []
    assert ols.select_from_similarities(sims, 4) == [3]
    assert ols.select_from_similarities(sims, 9) == [3]                      # standard: at most 1
    assert ols.select_from_similarities(sims, 9, max_replace_threshold=3) == [1, 3, 6]
    assert ols.select_from_similarities(sims, 3, use_distill=True) == [3]
    assert ols.select_from_similarities(sims, 4, use_distill=True) == [1, 3, 6]  # distill default: up to 3
    assert ols.select_from_similarities([0.5] * 8, 9) == [0]                  # nothing below the threshold -> the argmin


REFINE_CASES = {"spatial": dict(F0=5, H0=32, W0=48, H=64, W=128, steps=8, shift=1.0, t=0.5, sro=True),
                "spatiotemporal": dict(F0=3, H0=24, W0=40, H=64, W=64, steps=6, shift=4.0, t=0.6, sro=False)}


def refine_inputs(c):
    g = torch.Generator().manual_seed(23)
    frames = (torch.rand(c["F0"], c["H0"], c["W0"], 3, generator=g) * 255).to(torch.uint8)
    image = torch.rand(3, c["H"], c["W"], generator=g)
    pe = torch.randn(1, 1, 12, 32, generator=g).to(torch.bfloat16)
    pm = torch.zeros(1, 12, dtype=torch.int64)
    pm[:, :9] = 1
    return frames, image, pe, pm


@pytest.mark.parametrize("name,vae_dtype", [(n, torch.float32) for n in REFINE_CASES] + [("spatial", torch.bfloat16)])
def test_refine_pass_matches_reference(name, vae_dtype):
    """G15: the unmodified generate_refine (schedule truncation, bf16 upsampling chain, granularity padding, noise mixing, condition
    latents, Euler loop without CFG, frame slicing)."""
    c = REFINE_CASES[name]
    G = np.load(os.path.join(GOLD, f"g15_longcat_refine_{name}.npz" if vae_dtype == torch.float32 else f"g15b_longcat_refine_{name}_vaebf16.npz"))
    frames, image, pe, pm = refine_inputs(c)
    dit, vae = FakeLongCatDiT(), FakeVAE(vae_dtype)
    mean, std = vae.config.latents_mean, vae.config.latents_std
    sig, ts = ols.refine_schedule(c["steps"], c["shift"], c["t"])
    assert np.array_equal(sig.numpy(), G["sigmas"]) and np.array_equal(ts.numpy(), G["timesteps"])
    nf = c["F0"] if c["sro"] else 2 * c["F0"]
    up = ols.refine_upsample(frames, c["H"], c["W"], nf)
    ncl, ac, an, ncf = ols.refine_plan(nf, 1)
    assert (ncl, ac, ncf) == (4, 12, 13)
    assert np.array_equal(up.float().numpy(), G["video_up"][:, :, ac:ac + nf])
    trace = []
    out = ols.run_refine(stage1_frames=frames, image=(2.0 * image - 1.0)[None], height=c["H"], width=c["W"], dit=dit, prompt_embeds=pe,
                         prompt_mask=pm, encode_sample=lambda x, g: vae.encode(x).latent_dist.sample(g),
                         decode=lambda z: vae.decode(z)[0], mean=mean, std=std, generator=torch.manual_seed(42),
                         num_inference_steps=c["steps"], shift=c["shift"], t_thresh=c["t"], spatial_refine_only=c["sro"], trace=trace,
                         vae_dtype=vae_dtype)
    assert len(trace) == int(G["n"][0])
    for j, lat in enumerate(trace):
        np.testing.assert_allclose(lat[:, :, ncl:].numpy(), G[f"step{j}"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out.numpy(), G["frames"], rtol=0, atol=2e-5)
