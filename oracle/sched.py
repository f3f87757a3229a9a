"""Oracle: flow-matching UniPC(bh2) scheduler with the WorldForge resample extensions (torch CPU).

Restates /root/reference/wan_for_worldforge/utils/scheduling_unipc_multistep_clean.py (SCHED) for the configuration
the Wan checkpoints ship (prediction_type="flow_prediction", use_flow_sigmas=True, solver_order=2, bh2, predict_x0,
lower_order_final) -- the only branch the hot path reaches.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch


def flow_sigmas(num_steps: int, shift: float, num_train: int = 1000):
    """SCHED:812-818 -> (sigmas[N+1] float32 incl. trailing 0, timesteps[N] int64)."""
    alphas = np.linspace(1, 1 / num_train, num_steps + 1)
    sig = 1.0 - alphas
    sig = np.flip(shift * sig / (1 + (shift - 1) * sig))[:-1].copy()
    timesteps = (sig * num_train).copy()
    sig = np.concatenate([sig, [0]]).astype(np.float32)
    return torch.from_numpy(sig), torch.from_numpy(timesteps).to(torch.int64)


def resample_tables(sigmas: torch.Tensor, num_train: int = 1000):
    """SCHED:1594-1629 (flow branch): resample sigmas = sigmas[:-1]; timesteps = floor(sigma * num_train)."""
    rs = sigmas[:-1].clone()
    rt = torch.floor(rs * num_train).to(torch.int64)
    return rs, rt


def _lam(sigma: torch.Tensor):
    """SCHED:882-889, 1019: alpha = 1 - sigma; lambda = log(alpha) - log(sigma)."""
    alpha = 1 - sigma
    return alpha, sigma, torch.log(alpha) - torch.log(sigma)


@dataclass
class SchedState:
    sigmas: torch.Tensor
    timesteps: torch.Tensor
    resample_sigmas: torch.Tensor
    resample_timesteps: torch.Tensor
    solver_order: int = 2
    model_outputs: List[Optional[torch.Tensor]] = field(default_factory=lambda: [None, None])
    lower_order_nums: int = 0
    last_lower_order_nums: int = 0
    this_order: Optional[int] = None
    last_this_order: Optional[int] = None
    last_sample: Optional[torch.Tensor] = None
    step_index: Optional[int] = None
    is_resampling: bool = False
    original_step_index: Optional[int] = None
    derivative_history: list = field(default_factory=list)


def make_state(num_steps: int, shift: float) -> SchedState:
    """SCHED:769-846 set_timesteps."""
    sig, ts = flow_sigmas(num_steps, shift)
    rs, rt = resample_tables(sig)
    return SchedState(sigmas=sig, timesteps=ts, resample_sigmas=rs, resample_timesteps=rt)


def set_resample_mode(st: SchedState, enabled: bool):
    """SCHED:1631-1638."""
    if enabled and not st.is_resampling:
        st.original_step_index = st.step_index
    st.is_resampling = enabled
    if not enabled and st.original_step_index is not None:
        st.step_index = st.original_step_index
        st.original_step_index = None


def get_resample_timestep(st: SchedState, i: int) -> torch.Tensor:
    """SCHED:1640-1648."""
    if i < len(st.resample_timesteps):
        return st.resample_timesteps[i]
    return st.timesteps[min(i, len(st.timesteps) - 1)]


def index_for_timestep(st: SchedState, timestep, schedule=None) -> int:
    """SCHED:1224-1237."""
    schedule = st.timesteps if schedule is None else schedule
    cand = (schedule == timestep).nonzero()
    if len(cand) == 0:
        return len(st.timesteps) - 1
    if len(cand) > 1:
        return cand[1].item()
    return cand[0].item()


def current_sigma(st: SchedState) -> torch.Tensor:
    """SCHED:953-957: sigma used by convert_model_output."""
    if st.is_resampling:
        return st.resample_sigmas[min(st.step_index, len(st.resample_sigmas) - 1)]
    return st.sigmas[st.step_index]


def convert_model_output(st: SchedState, v: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
    """SCHED:958: x0 = sample - sigma_t * v   (0-dim fp32 sigma: the product keeps v's dtype)."""
    return sample - current_sigma(st) * v


def unip_coeffs(st: SchedState, order: int):
    """SCHED:1005-1069 scalar part: returns (c1, c2, c3, rk) as 0-dim fp32 tensors.
    x_t = c1*x - c2*m0 - c3 * 0.5 * (m1 - m0)/rk."""
    if st.is_resampling:
        n = len(st.resample_sigmas)
        cur = min(st.step_index, n - 1)
        nxt = min(st.step_index + 1, n - 1)
        sigma_t = st.sigmas[nxt]  # SCHED:1008-1009 (next_index < n always holds)
        sigma_s0 = st.resample_sigmas[cur]
    else:
        sigma_t, sigma_s0 = st.sigmas[st.step_index + 1], st.sigmas[st.step_index]
    alpha_t, sigma_t, lambda_t = _lam(sigma_t)
    alpha_s0, sigma_s0, lambda_s0 = _lam(sigma_s0)
    h = lambda_t - lambda_s0
    rk = None
    if order == 2:
        si = st.step_index - 1
        if st.is_resampling:
            sig_si = st.resample_sigmas[min(max(si, 0), len(st.resample_sigmas) - 1)]
        else:
            sig_si = st.sigmas[si]
        _, _, lambda_si = _lam(sig_si)
        rk = (lambda_si - lambda_s0) / h
    hh = -h
    h_phi_1 = torch.expm1(hh)
    B_h = torch.expm1(hh)  # bh2
    c1 = sigma_t / sigma_s0
    c2 = alpha_t * h_phi_1
    c3 = alpha_t * B_h
    return c1, c2, c3, rk


def unip_update(st: SchedState, sample: torch.Tensor, order: int) -> torch.Tensor:
    """SCHED:1083-1098 tensor part (predict_x0)."""
    c1, c2, c3, rk = unip_coeffs(st, order)
    m0 = st.model_outputs[-1]
    x = sample
    x_t_ = c1 * x - c2 * m0
    if order == 2:
        m1 = st.model_outputs[-2]
        D1 = (m1 - m0) / rk
        rhos_p = torch.tensor([0.5], dtype=x.dtype)
        pred_res = torch.einsum("k,bkc...->bc...", rhos_p, torch.stack([D1], dim=1))
        x_t = x_t_ - c3 * pred_res
    else:
        x_t = x_t_ - c3 * 0
    return x_t.to(x.dtype)


def step(st: SchedState, v: torch.Tensor, timestep, sample: torch.Tensor, *, fuse=None, resampling: bool = False,
         is_resample_round: bool = False, resample_count: int = 2):
    """SCHED:1423-1536.  `fuse` is a callable x0 -> x0 (the guided injection) or None.  Returns (prev_sample, pred_x0)."""
    if st.step_index is None:
        st.step_index = index_for_timestep(st, timestep)
    use_corrector = st.step_index > 0 and st.last_sample is not None  # disable_corrector == []
    x0 = convert_model_output(st, v, sample)
    if fuse is not None:
        x0 = fuse(x0)
    if not resampling:
        for i in range(st.solver_order - 1):
            st.model_outputs[i] = st.model_outputs[i + 1]
    st.model_outputs[-1] = x0
    this_order = min(st.solver_order, len(st.timesteps) - st.step_index)  # lower_order_final
    st.last_this_order = st.this_order
    st.this_order = min(this_order, st.lower_order_nums + 1)
    assert st.this_order > 0
    if not use_corrector:
        st.last_sample = sample
    if not is_resample_round:
        st.last_sample = sample
    if resample_count < 2:
        st.last_sample = sample
    if resampling:
        st.derivative_history.append(v)
    prev = unip_update(st, sample, st.this_order)
    st.last_lower_order_nums = st.lower_order_nums
    if st.lower_order_nums < st.solver_order:
        st.lower_order_nums += 1
    st.step_index += 1
    return prev, x0


def add_noise_resample(st: SchedState, x0: torch.Tensor, noise: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
    """SCHED:1542-1585 with use_resample_sigma=True and begin_index None: k = index of `timestep` in resample_timesteps."""
    sig = st.resample_sigmas.to(dtype=x0.dtype)
    k = index_for_timestep(st, timestep.reshape(-1)[0], st.resample_timesteps)
    s = sig[[k]].flatten()
    while s.dim() < x0.dim():
        s = s.unsqueeze(-1)
    return (1 - s) * x0 + s * noise
