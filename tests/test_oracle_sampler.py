"""CPU: the oracle's sampler / scheduler / injection restatement against goldens recorded from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import inject, sampler, sched
from tests.cases import PIPE_CASES, case_inputs
from tests.fakes import VAE_MEAN, VAE_STD, FakeDiT, FakeVAE


def test_schedule_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_schedules.npz"))
    for n in (4, 16, 50):
        for shift in (3.0, 5.0):
            st = sched.make_state(n, shift)
            k = f"n{n}_s{int(shift)}"
            assert np.array_equal(st.timesteps.numpy(), g[k + "_timesteps"])
            assert np.array_equal(st.sigmas.numpy(), g[k + "_sigmas"])
            assert np.array_equal(st.resample_sigmas.numpy(), g[k + "_rsig"])
            assert np.array_equal(st.resample_timesteps.numpy(), g[k + "_rts"])
    st = sched.make_state(4, 3.0)
    assert st.timesteps.tolist() == [999, 899, 749, 499]


def run_oracle_case(c):
    image, ref, mask, pe, ne, ie = case_inputs(c)
    dit, vae = FakeDiT(), FakeVAE()
    gen = torch.manual_seed(42)
    T = (c["F"] - 1) // 4 + 1
    latents = torch.randn((1, 16, T, c["H"] // 8, c["W"] // 8), generator=gen, dtype=torch.float32)
    dec = lambda z: vae.decode(z)[0]
    enc = lambda x: vae.encode(x).latent_dist.mode()
    cond = sampler.prepare_condition((2.0 * image - 1.0).unsqueeze(0), c["F"], enc, VAE_MEAN, VAE_STD)
    cfg = sampler.SamplerConfig(num_inference_steps=c["steps"], guidance_scale=c["cfg"], flow_shift=c["shift"],
                                guided=c["guided"], resample_steps=c["R"], guide_steps=c["guide"], omega=c["omega"],
                                omega_resample=c["omega_r"], resample_round=c["rnd"],
                                use_pca_channel_selection=c["flf"])
    trace = []
    tr = lambda x, t, ctx, img: dit(x, t, ctx, img)[0]
    lat = sampler.run(cfg, latents=latents, condition=cond, transformer=tr, prompt_embeds=pe, negative_prompt_embeds=ne,
                      image_embeds=ie, video_ref=ref, mask=mask, decode=dec, encode_mode=enc, mean=VAE_MEAN, std=VAE_STD,
                      generator=gen, trace=trace)
    frames = sampler.decode_final(lat, dec, VAE_MEAN, VAE_STD)
    return trace, lat, frames, (dit.calls, vae.n_enc, vae.n_dec)


@pytest.mark.parametrize("name", list(PIPE_CASES))
def test_sampler_matches_reference(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"g6_pipe_{name}.npz"))
    trace, lat, frames, counts = run_oracle_case(PIPE_CASES[name])
    assert list(counts) == g["n_calls"].tolist()
    steps = [t for t in trace if t[0] == "step"]
    lats = [t for t in trace if t[0] == "latents"]
    assert len(steps) == int(g["n_step_calls"][0])
    for j, (_, i, r, prev, x0) in enumerate(steps):
        assert str(prev.dtype) == g[f"call{j}_dtypes"][0] and str(x0.dtype) == g[f"call{j}_dtypes"][1], (j, i, r)
        np.testing.assert_array_equal(prev.float().numpy(), g[f"call{j}_prev"], err_msg=f"prev call {j} (i={i}, r={r})")
        np.testing.assert_array_equal(x0.float().numpy(), g[f"call{j}_x0"], err_msg=f"x0 call {j} (i={i}, r={r})")
    for j, (_, i, l) in enumerate(lats):
        assert str(l.dtype) == g[f"lat{j}_dtype"][0]
        np.testing.assert_array_equal(l.float().numpy(), g[f"lat{j}"], err_msg=f"latents after step {i}")
    np.testing.assert_array_equal(frames.numpy(), g["frames"])


def test_flow_metric_and_selection(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_flf.npz"))
    for k in range(5):
        r, c = torch.from_numpy(g[f"fm{k}_ref"]), torch.from_numpy(g[f"fm{k}_chan"])
        assert inject.flow_similarity(r, c) == pytest.approx(float(g[f"fm{k}_sim"][0]), abs=0, rel=0)
    for si in range(5):
        sims = g[f"sel{si}_sims"]
        for step in (0, 1, 2, 3, 5, 6, 10, 11, 30):
            assert inject.select_from_similarities(sims, step) == g[f"sel{si}_step{step}"].tolist(), (si, step)
    pred, enc = torch.from_numpy(g["e2e_pred"]), torch.from_numpy(g["e2e_enc"])
    np.testing.assert_array_equal(np.array(inject.channel_similarities(pred, enc)), g["e2e_sims"])
    for step in (3, 8, 12):
        assert inject.select_motion_related_channels(pred, enc, step) == g[f"e2e_step{step}"].tolist()


def test_frame_count_mismatch_raises_like_reference():
    ref = torch.rand(1, 3, 7, 16, 16)
    mask = torch.ones(1, 1, 7, 16, 16)
    with pytest.raises(ValueError):
        inject.align_reference(ref, mask, (1, 3, 5, 16, 16))
