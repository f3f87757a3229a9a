"""GPU: the batched Farneback flow (csrc/flow.hip) against the CPU restatement oracle/farneback.py (parity with a real cv2 is
UNPINNED for both -- see the oracle header), and the FLF gate on top of it."""
import numpy as np
import pytest
import torch

from oracle import farneback as ofb
from oracle import inject as oinject

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _latents(C, T, h, w, seed, smooth=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(C, T, h, w, generator=g)
    if smooth:  # latents are spatially smooth and drift over time; pure noise has no trackable structure
        k = torch.ones(1, 1, 5, 5) / 25.0
        x = torch.nn.functional.conv2d(x.reshape(C * T, 1, h, w), k, padding=2).reshape(C, T, h, w) * 3
        x = x + torch.cumsum(torch.randn(C, T, 1, 1, generator=g) * 0.1, dim=1)
    return x.contiguous()


@pytest.mark.parametrize("C,T,h,w,dtype", [
    (3, 4, 60, 104, torch.float32),     # the C2 latent grid: one pyramid level
    (2, 3, 90, 160, torch.float32),     # C3: two levels (pre-blur sigma .5, 2x resize, flow upsampling)
    (1, 3, 128, 256, torch.float32),    # three levels
    (2, 5, 20, 24, torch.float32),
    (2, 3, 7, 9, torch.float32),        # narrower than 2 x BORDER: the unsigned edge test of UpdateMatrices
    (2, 3, 4, 4, torch.bfloat16),       # the latent size of the tiny sampler cases
    (2, 4, 58, 104, torch.bfloat16),    # C1
])
def test_farneback_flows_match_oracle(C, T, h, w, dtype):
    from worldforge_amd import ops
    x = _latents(C, T, h, w, seed=h * 7 + w).to(dtype)
    got = ops.farneback_flows(x.to(DEV)).cpu().numpy()
    x32 = x.float().numpy()
    gmin = np.float32(x32.min())
    grange = np.float32(np.float32(x32.max()) - gmin + np.float32(1e-8))
    assert got.shape == (C, T - 1, 2, h, w)
    for c in range(C):
        want = ofb.channel_flow(x32[c], gmin, grange)
        err = np.abs(got[c] - want).max()
        scale = max(1.0, np.abs(want).max())
        assert err <= 2e-3 * scale, (c, err, scale)


def test_quantisation_is_bit_exact():
    """The uint8 grey levels feed a discontinuous stage (truncation): they must be identical.  Checked through a case where the flow
    is driven only by the quantised values: two tensors that differ by less than a grey level step must give identical flows."""
    from worldforge_amd import ops
    x = _latents(2, 3, 32, 40, seed=5)
    a = ops.farneback_flows(x.to(DEV))
    b = ops.farneback_flows(x.to(DEV).clone())
    assert torch.equal(a, b)                                     # deterministic
    x32 = x.numpy()
    gmin = np.float32(x32.min())
    grange = np.float32(np.float32(x32.max()) - gmin + np.float32(1e-8))
    q = ofb.quantise_channel(x32[0], gmin, grange)
    assert q.min() >= 0 and q.max() <= 255 and q.dtype == np.uint8


@pytest.mark.parametrize("step", [3, 8, 20])
def test_flf_gate_with_farneback_backend_matches_oracle(step):
    from worldforge_amd.flf import VideoMotionPCASelector
    enc = _latents(16, 5, 24, 32, seed=11).unsqueeze(0)
    pred = (enc + 0.3 * _latents(16, 5, 24, 32, seed=12).unsqueeze(0)).to(torch.bfloat16)
    sel = VideoMotionPCASelector(flow_backend="farneback")
    chans = sel.select_motion_related_channels(pred.to(DEV), enc.to(DEV), current_step=step)
    want_s = np.array(oinject.channel_similarities(pred, enc, flow_backend="farneback"))
    assert np.abs(sel.last_similarities - want_s).max() <= 2e-3
    want = oinject.select_from_similarities(sel.last_similarities, step)
    assert chans == want


@pytest.mark.parametrize("C,T,h,w,dtype", [
    (16, 4, 30, 52, torch.float32),     # the LongCat 480p latent grid
    (3, 3, 90, 160, torch.bfloat16),    # 720p refine grid, two pyramid levels, normalisation in bf16
    (2, 3, 4, 4, torch.bfloat16),
    (16, 3, 8, 8, torch.float32),
])
def test_longcat_quantisation_mode_matches_oracle(C, T, h, w, dtype):
    """quant_mode 1: a range per channel, normalisation in the tensor's dtype, uint8(clip((n + 1) * 127.5)) -- LongCat scheduler :105-121,
    290-297 as restated by oracle.longcat_sampler.farneback_motion."""
    from oracle import longcat_sampler as ols
    from worldforge_amd import ops
    x = _latents(C, T, h, w, seed=h * 5 + w).to(dtype)
    x[0] = x[0] * 7 + 3                  # ranges differ per channel: a global range would give other grey levels
    got = ops.farneback_flows(x.to(DEV), quant_mode=1).cpu().numpy()
    for c in range(C):
        want = ols.farneback_motion(x[None, c:c + 1])[0].numpy()
        err = np.abs(got[c] - want).max()
        scale = max(1.0, np.abs(want).max())
        assert err <= 2e-3 * scale, (c, err, scale)
    if h >= 24:
        assert not np.array_equal(got, ops.farneback_flows(x.to(DEV), quant_mode=0).cpu().numpy())


def test_longcat_constant_channel_is_all_zero_grey():
    """max == min: the range is 1e-8, every value normalises to 0 -> grey 127 everywhere -> zero flow (no NaN)."""
    from worldforge_amd import ops
    x = _latents(2, 3, 16, 16, seed=3)
    x[1] = 0.25
    got = ops.farneback_flows(x.to(DEV), quant_mode=1)
    assert torch.isfinite(got).all() and got[1].abs().max().item() == 0.0


@pytest.mark.parametrize("step,distill", [(3, False), (8, False), (8, True)])
def test_longcat_flf_gate_with_farneback_backend_matches_oracle(step, distill):
    from oracle import longcat_sampler as ols
    from worldforge_amd.longcat_scheduler import VideoMotionChannelSelector
    enc = _latents(16, 4, 24, 32, seed=21).unsqueeze(0)
    pred = enc + 0.3 * _latents(16, 4, 24, 32, seed=22).unsqueeze(0)
    sel = VideoMotionChannelSelector()
    assert sel.flow_backend == "farneback" and sel.use_optical_flow
    chans = sel.select_motion_related_channels(pred.to(DEV), enc.to(DEV), current_step=step, use_distill=distill)
    want_s = np.array(ols.channel_similarities(pred, enc, flow_backend="farneback"))
    assert np.abs(sel.last_similarities - want_s).max() <= 2e-3
    assert chans == ols.select_from_similarities(sel.last_similarities, step, distill)
    with pytest.raises(ValueError):
        VideoMotionChannelSelector("cv2")
