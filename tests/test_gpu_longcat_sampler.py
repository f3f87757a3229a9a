"""GPU: the HIP-backed LongCat guided sampler (worldforge_amd/longcat_pipeline.py + longcat_scheduler.py) against the trajectories
recorded from the unmodified reference pipeline (tests/golden/g12_longcat_pipe_*.npz) with the deterministic DiT / VAE stand-ins, and
its reduction kernels against the CPU oracle.

Tolerances: the stand-ins are element-wise IEEE arithmetic in a fixed order, and the product's latent kernels round every operation
separately (-ffp-contract=off), so trajectories agree to fp32 rounding of the two global reductions (CFG-zero, DSG: a fixed tree on the
GPU vs torch's order on the CPU): |err| <= 2e-4 on O(1) latents after up to 25 scheduler steps, frames <= 1e-3."""
import os

import numpy as np
import pytest
import torch

from oracle import inject as oinj
from oracle import longcat_sampler as ols
from tests.fakes import FakeLongCatDiT, FakeVAE
from tests.test_oracle_longcat_sampler import CASES, GOLD, case_inputs

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("n,g", [(16 * 3 * 8 * 12, 4.0), (16 * 24 * 60 * 104, 3.0), (7, 1.5)])
def test_cfg_zero_kernel(n, g):
    from worldforge_amd import ops
    c, u = _rand((n,), 1), _rand((n,), 2) * 0.7 + 0.3 * _rand((n,), 1)
    want = -ols.cfg_zero(c.view(1, 1, 1, 1, n).double(), u.view(1, 1, 1, 1, n).double(), g).view(-1)
    got = ops.cfg_zero(c.to(DEV), u.to(DEV), g, negate=True).cpu()
    assert (got.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    pos = ops.cfg_zero(c.to(DEV), u.to(DEV), g).cpu()
    assert torch.equal(pos, -got)


@pytest.mark.parametrize("scale_r,scale_c", [(1.0, 1.0), (6.0, 5.0), (0.05, 0.08), (12.0, 12.0)])
def test_flow_metric_longcat_variant(scale_r, scale_c):
    from worldforge_amd import ops
    r = _rand((5, 6, 1, 10, 12), 3, scale_r)
    c = r * 0.7 + _rand((5, 6, 1, 10, 12), 4, scale_c)
    got = ops.flow_metrics(r.to(DEV), c.to(DEV), variant=1).cpu()
    want = torch.tensor([ols.flow_similarity(r[i:i + 1], c[i:i + 1]) for i in range(5)])
    assert (got - want).abs().max().item() <= 2e-5
    wan = ops.flow_metrics(r.to(DEV), c.to(DEV), variant=0).cpu()
    want0 = torch.tensor([oinj.flow_similarity(r[i:i + 1], c[i:i + 1]) for i in range(5)])
    assert (wan - want0).abs().max().item() <= 2e-5


def test_scheduler_schedule_equals_reference():
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    for name, c in CASES.items():
        G = np.load(os.path.join(GOLD, f"g12_longcat_pipe_{name}.npz"))
        sch = FlowMatchEulerDiscreteScheduler(shift=c["shift"])
        pipe = LongCatVideoPipeline(FakeVAE(), sch, FakeLongCatDiT(), device=DEV)
        sch.set_timesteps(c["steps"], sigmas=pipe.get_timesteps_sigmas(c["steps"], c["distill"]))
        assert np.array_equal(sch.sigmas.numpy(), G["sigmas"]) and np.array_equal(sch.timesteps.numpy(), G["timesteps"])
    with pytest.raises(NotImplementedError):
        FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=True)
    with pytest.raises(ValueError):
        sch.step(torch.zeros(1, device=DEV), 3, torch.zeros(1, device=DEV))


@pytest.mark.parametrize("name", list(CASES))
def test_pipeline_matches_reference_trajectory(name):
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    c = CASES[name]
    G = np.load(os.path.join(GOLD, f"g12_longcat_pipe_{name}.npz"))
    image, ref, mask, pe, pm, ne, nm = case_inputs(c)
    dit, vae = FakeLongCatDiT(), FakeVAE()
    sch = FlowMatchEulerDiscreteScheduler(shift=c["shift"])
    pipe = LongCatVideoPipeline(vae, sch, dit, device=DEV)
    calls = []
    orig_step = sch.step

    def wrapped(*a, **k):
        o = orig_step(*a, **k)
        calls.append((o.prev_sample.cpu().numpy(), o.pred_x0.cpu().numpy()))
        return o

    sch.step = wrapped
    lat0 = {}
    orig_prep = pipe.prepare_latents

    def prep(*a, **k):
        lat = orig_prep(*a, **k)
        lat0["v"] = lat.cpu().numpy().copy()
        return lat

    pipe.prepare_latents = prep
    frames = pipe.generate_i2v(image=image, height=c["H"], width=c["W"], prompt_embeds=pe, prompt_attention_mask=pm,
                               negative_prompt_embeds=ne, negative_prompt_attention_mask=nm, num_frames=c["F"],
                               num_inference_steps=c["steps"], use_distill=c["distill"], guidance_scale=c["cfg"],
                               generator=torch.manual_seed(42), output_type="np", video_ref=ref, mask=mask, guided=c["guided"],
                               resample_steps=c["R"], guide_steps=c["guide"], resample_round=c["rnd"], omega=c["omega"],
                               omega_resample=c["omega_r"], use_pca_channel_selection=c["flf"], static=True,
                               max_replace_threshold=c["maxrep"])
    np.testing.assert_allclose(lat0["v"], G["latents0"], rtol=0, atol=1e-6)
    assert len(calls) == int(G["n_step_calls"][0])
    assert [dit.calls, vae.n_enc, vae.n_dec] == G["n_calls"].tolist()
    for j, (prev, x0) in enumerate(calls):
        np.testing.assert_allclose(prev, G[f"call{j}_prev"], rtol=0, atol=2e-4, err_msg=f"{name} call {j} prev")
        np.testing.assert_allclose(x0, G[f"call{j}_x0"], rtol=0, atol=2e-4, err_msg=f"{name} call {j} x0")
    np.testing.assert_allclose(frames, G["frames"], rtol=0, atol=1e-3)
