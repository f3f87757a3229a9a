"""GPU: the HIP Farneback kernels (csrc/flow.hip, `wf_farneback_flows`) against what a REAL cv2.calcOpticalFlowFarneback returned for the
same uint8 frames (tests/golden/g20_farneback_cv2.npz, tools/record_thirdparty_goldens.py).  Skips until that file has been recorded on a
machine with opencv-python (absent from /root/reference and from the build image).  The crack-fill and point-render kernels are compared
with their oracles in test_gpu_warp.py; the oracles meet the packages in tests/test_thirdparty_goldens.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "g20_farneback_cv2.npz")


def _hip_flows_of_uint8_frames(frames):
    from worldforge_amd import ops
    C, T, h, w = frames.shape
    # wf_farneback_flows quantises itself: uint8(255 * (x - min) / (max - min)) over the WHOLE tensor (SCHED:376-388, 175).  Feed it values
    # that land in the middle of the recorded grey level's bin: an extra calibration channel pins min = 0 and max = 256, so a real channel
    # holding (v + 0.5) * 256 / 255 quantises to exactly v.
    x = np.empty((C + 1, T, h, w), dtype=np.float32)
    x[:C] = (frames.astype(np.float32) + 0.5) * np.float32(256.0 / 255.0)
    x[C] = 128.0
    x[C, :, 0, 0], x[C, :, 0, 1] = 0.0, 256.0
    return ops.farneback_flows(torch.from_numpy(x).to("cuda:0"), quant_mode=0)[:C].cpu().numpy()


@pytest.mark.parametrize("name", ["latent", "odd"])
def test_hip_farneback_equals_cv2(name):
    if not os.path.exists(GOLD):
        pytest.skip("g20_farneback_cv2.npz not recorded yet: run tools/record_thirdparty_goldens.py where opencv-python is installed")
    g = np.load(GOLD)
    frames, want = g[f"{name}_frames"], g[f"{name}_flows"]                  # uint8 [C, T, h, w], f32 [C, T-1, 2, h, w]
    err = float(np.abs(_hip_flows_of_uint8_frames(frames) - want).max())
    assert err <= 3e-3, err


def test_uint8_feeding_reproduces_the_frames_exactly_vs_oracle():
    """Always runs: the calibration-channel feeding above hands the kernel exactly the recorded grey levels -- checked against the oracle's
    flows of the same uint8 frames (so that the cv2 comparison, once it can run, compares like with like)."""
    from oracle import farneback as ofb
    from tests import thirdparty_cases as tc
    frames = tc.farneback_frames("odd")
    C, T = frames.shape[:2]
    want = np.stack([np.stack([ofb.calc_optical_flow_farneback(frames[c, t], frames[c, t + 1]).transpose(2, 0, 1) for t in range(T - 1)])
                     for c in range(C)])
    assert float(np.abs(_hip_flows_of_uint8_frames(frames) - want).max()) <= 2e-3
