"""Shared description of the golden sampler cases (must match tools/make_goldens.py PIPE_CASES)."""
import torch

from tests.fakes import synthetic_ref_and_mask

PIPE_CASES = {
    "irr_dsg_small": dict(steps=4, R=2, guide=3, rnd=3, flf=False, omega=4.0, omega_r=4.0, cfg=4.0, F=9, H=32, W=32,
                          guided=True, shift=3.0),
    "flf_full": dict(steps=14, R=2, guide=12, rnd=12, flf=True, omega=4.0, omega_r=2.0, cfg=4.0, F=9, H=32, W=48,
                     guided=True, shift=3.0),
    "guide_lt_round": dict(steps=10, R=2, guide=4, rnd=7, flf=True, omega=6.0, omega_r=1.5, cfg=5.0, F=5, H=32, W=32,
                           guided=True, shift=5.0),
    "plain": dict(steps=6, R=1, guide=0, rnd=0, flf=False, omega=1.8, omega_r=1.0, cfg=5.0, F=5, H=32, W=32,
                  guided=False, shift=3.0),
    "nocfg_R3": dict(steps=5, R=3, guide=3, rnd=4, flf=False, omega=4.0, omega_r=4.0, cfg=1.0, F=5, H=32, W=32,
                     guided=True, shift=3.0),
}


def case_inputs(c, seed=42):
    g = torch.Generator().manual_seed(1000 + seed)
    image = torch.rand(3, c["H"], c["W"], generator=g)
    ref, mask = synthetic_ref_and_mask(c["F"], c["H"], c["W"], seed=seed)
    ref[:, :, 0] = image
    pe = torch.randn(1, 16, 32, generator=g).to(torch.bfloat16)
    ne = torch.randn(1, 16, 32, generator=g).to(torch.bfloat16)
    ie = torch.randn(1, 8, 16, generator=g)
    return image, ref, mask, pe, ne, ie


# fill_small_cracks cases of tests/golden/g23_fill_small_cracks.npz (tools/make_goldens.py small_cracks): name -> (a confidence map exists,
# max_crack_size, min_valid_neighbors, depth_threshold).  With the shipped min_valid_neighbors (3; run_warp.py passes 2) step 1 fills every
# hole step 2 could reach; the stricter counts make the depth-guided sequential step do work.
SMALL_CRACK_CASES = {
    "conf": (True, 5, 3, 0.1), "noconf": (False, 5, 3, 0.1), "conf_mvn2": (True, 6, 2, 0.1), "conf_mvn6_small": (True, 2, 6, 0.1),
    "noconf_mvn7": (False, 5, 7, 0.1), "conf_mvn7": (True, 5, 7, 0.1), "conf_mvn7_thr1": (True, 5, 7, 1.0), "conf_mvn7_size1": (True, 1, 7, 1.0),
}
