"""Deterministic stand-ins for the DiT and the VAE, used to pin the *sampler state machine* (scheduler, IRR, FLF, DSG).

They are built only from element-wise IEEE multiplies/adds and index operations in a fixed order, so the same inputs
give bit-identical outputs on CPU and on a GPU -- which lets the golden trajectories recorded from the reference
pipeline (tools/make_goldens.py) be compared tightly with both the CPU oracle and the HIP-backed product sampler.
They follow the diffusers call protocol the reference uses (PIPE:593-600, SCHED:1272-1285, 1384).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
            -0.1922, -0.9497, 0.2503, -0.2921]
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
           1.1253, 2.8251, 1.9160]


def _coef(n, m, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, m, generator=g) * 2 - 1) * scale


class FakeDiT:
    """v[c] = a_c(t) * x[c] + b_c * x[16 + c] + g_c * x[(c + 5) % 16] + bias_c(ctx)   -> transformer dtype."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype
        self.config = SimpleNamespace(patch_size=(1, 2, 2))
        self.coef = _coef(16, 3, 1234, 0.8)
        self.calls = 0

    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
                 attention_kwargs=None, return_dict=False):
        self.calls += 1
        x = hidden_states.float()
        ts = (timestep.reshape(-1)[0].float() / 1000.0).item()
        ctx = encoder_hidden_states.float()[0, 0, :4].cpu().tolist()
        outs = []
        for c in range(16):
            a, b, g = [float(v) for v in self.coef[c]]
            bias = ctx[c % 4] * 0.25
            o = x[:, c] * (a * (0.5 + ts)) + x[:, 16 + c] * b
            o = o + x[:, (c + 5) % 16] * g
            o = o + bias
            outs.append(o)
        return (torch.stack(outs, dim=1).to(self.dtype),)


class _Dist:
    def __init__(self, mu):
        self._mu = mu

    def mode(self):
        return self._mu

    def sample(self, generator=None):
        """Posterior sample with a fixed small std (LongCat's prepare_latents draws it: pipeline_longcat_video.py:278)."""
        # drawn in the distribution's dtype, as diffusers' DiagonalGaussianDistribution.sample does (randn_tensor(..., dtype=parameters.dtype))
        noise = torch.randn(self._mu.shape, generator=generator, dtype=self._mu.dtype)
        return self._mu + noise.to(self._mu.device) * 0.05


class FakeLongCatDiT:
    """LongCat call protocol (pipeline_longcat_video.py:867-873): batch of CFG samples, per-frame timesteps, caption masks.
    v[b,c] = a_c * (0.5 + t[b,frame]/1000) * x[b,c] + g_c * x[b,(c+5)%16] + 0.25 * ctx[b,c%4] + 0.01 * n_valid[b]  -> fp32."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype
        self.config = SimpleNamespace(in_channels=16, out_channels=16, patch_size=(1, 2, 2))
        self.cp_split_hw = None
        self.coef = _coef(16, 3, 4321, 0.8)
        self.calls = 0

    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None, num_cond_latents=0, **kw):
        self.calls += 1
        x = hidden_states.float()
        B, _, T = x.shape[:3]
        ts = timestep.float()
        if ts.dim() == 1:
            ts = ts.unsqueeze(1).expand(-1, T)
        fac = (ts.to(x.device) / 1000.0 + 0.5).view(B, T, 1, 1)
        ctx = encoder_hidden_states.float().reshape(B, -1, encoder_hidden_states.shape[-1])[:, 0, :4].cpu().tolist()
        nval = [float(v) for v in encoder_attention_mask.sum(dim=-1).reshape(-1).cpu().tolist()] if encoder_attention_mask is not None else [0.0] * B
        outs = []
        for c in range(16):
            a, _, g = [float(v) for v in self.coef[c]]
            o = x[:, c] * fac * a
            o = o + x[:, (c + 5) % 16] * g
            bias = torch.tensor([ctx[b][c % 4] * 0.25 + 0.01 * nval[b] for b in range(B)], dtype=torch.float32, device=x.device)
            o = o + bias.view(B, 1, 1, 1)
            outs.append(o)
        return torch.stack(outs, dim=1)


class FakeVAE:
    """decode: channel mix 16->3, nearest x8 spatial, latent frame t -> output frames (1 + 4(T-1)), clamp(-1,1);
    encode: frames 0,4,8,.., pixels ::8, channel mix 3->16."""

    def __init__(self, dtype=torch.float32):
        # dtype = the module dtype (`vae.dtype`): torch.bfloat16 is how the LongCat entry loads its VAE
        # (run_longcat_worldforge_single.py:205).  Like a module in that dtype, the stand-in then refuses inputs of another dtype, computes
        # every element-wise step in it (one rounding per step) and returns it.
        self.dtype = dtype
        self.config = SimpleNamespace(z_dim=16, latents_mean=VAE_MEAN, latents_std=VAE_STD, scale_factor_temporal=4, scale_factor_spatial=8)
        self.temperal_downsample = [False, True, True]
        self.wd = _coef(3, 16, 77, 0.3)
        self.we = _coef(16, 3, 78, 1.2)
        self.n_dec = 0
        self.n_enc = 0

    def decode(self, z, return_dict=False):
        self.n_dec += 1
        if self.dtype != torch.float32 and z.dtype != self.dtype:
            raise RuntimeError(f"FakeVAE({self.dtype}).decode: input is {z.dtype} (a module in {self.dtype} refuses it)")
        B, C, T, h, w = z.shape
        chans = []
        for o in range(3):
            acc = z[:, 0] * float(self.wd[o, 0])
            for k in range(1, 16):
                acc = acc + z[:, k] * float(self.wd[o, k])
            chans.append(acc)
        x = torch.stack(chans, dim=1)  # [B,3,T,h,w]
        idx = torch.tensor([0] + [1 + (f - 1) // 4 for f in range(1, 1 + 4 * (T - 1))], device=z.device)
        x = x.index_select(2, idx)
        x = x.repeat_interleave(8, dim=3).repeat_interleave(8, dim=4)
        return (x.clamp(-1, 1),)

    def encode(self, x):
        self.n_enc += 1
        if self.dtype != torch.float32 and x.dtype != self.dtype:
            raise RuntimeError(f"FakeVAE({self.dtype}).encode: input is {x.dtype} (a module in {self.dtype} refuses it)")
        s = x.to(self.dtype)[:, :, ::4, ::8, ::8]
        chans = []
        for o in range(16):
            acc = s[:, 0] * float(self.we[o, 0])
            for k in range(1, 3):
                acc = acc + s[:, k] * float(self.we[o, k])
            chans.append(acc)
        return SimpleNamespace(latent_dist=_Dist(torch.stack(chans, dim=1)))


def synthetic_ref_and_mask(F, H, W, seed=5, soften=True):
    """Reference video [1,3,F,H,W] in [0,1] and a growing-hole mask [1,1,F,H,W]."""
    g = torch.Generator().manual_seed(seed)
    ref = torch.rand(1, 3, F, H, W, generator=g)
    xs = torch.arange(W).view(1, 1, 1, 1, W).float()
    fr = torch.arange(F).view(1, 1, F, 1, 1).float() / max(F - 1, 1)
    edge = W * (1 - 0.35 * fr)
    mask = (xs < edge).float().expand(1, 1, F, H, W).contiguous()
    if soften:
        d = (edge - xs).clamp(min=0)
        soft = torch.sin(torch.pi / 2 * (d / 4).clamp(0, 1))
        mask = (mask * soft.expand_as(mask)).contiguous()
    return ref, mask


class SimComm:
    """Stand-in for parallel.Comm that runs P simulated ranks as threads of ONE process on one GPU: all_gather deposits each
    rank's tensor in a shared slot and meets at a barrier.  Exercises the row-slab sharded VAE (halo exchange, slab
    convolutions, row gather) without a multi-GPU node."""

    def __init__(self, world, rank, shared):
        self.world, self.rank, self.sh = world, rank, shared

    def split(self, n_groups):
        """parallel.Comm.split: contiguous sub-groups; the threads of a sub-group meet in their own shared slot table."""
        import threading
        per = self.world // n_groups
        g = self.rank // per
        sub_sh = self.sh.setdefault(("sub", n_groups, g), {"slots": [None] * per, "bar": threading.Barrier(per)})   # (atomic in CPython)
        sub = SimComm(per, self.rank % per, sub_sh)
        sub.group_index = g
        return sub

    def all_gather(self, out, inp):
        sh = self.sh
        sh["slots"][self.rank] = inp
        torch.cuda.current_stream().synchronize()   # the producers of `inp` ran on this thread's stream
        sh["bar"].wait()
        for r in range(self.world):
            if r != self.rank or inp.data_ptr() != out[r].data_ptr():   # (in-place form of parallel.KVExchange: the own slot stays)
                out[r].copy_(sh["slots"][r])
        torch.cuda.current_stream().synchronize()
        sh["bar"].wait()
        return out

    def all_gather_async(self, out, inp):
        self.all_gather(out, inp)
        return None

    def neighbor_rows(self, top, bottom, d=1):
        """parallel.Comm.neighbor_rows: (bottom row of rank - d, top row of rank + d), None at the ends."""
        sh = self.sh
        sh["slots"][self.rank] = (top.clone(), bottom.clone())
        torch.cuda.current_stream().synchronize()
        sh["bar"].wait()
        up = sh["slots"][self.rank - d][1].clone() if self.rank - d >= 0 else None
        down = sh["slots"][self.rank + d][0].clone() if self.rank + d < self.world else None
        torch.cuda.current_stream().synchronize()
        sh["bar"].wait()
        return up, down

    def neighbor_rows_async(self, top, bottom, d=1):
        res = self.neighbor_rows(top, bottom, d)
        return lambda: res

    def broadcast_slots_async(self, out):
        """parallel.Comm.broadcast_slots_async: slot `rank` of `out` is this rank's data; the other slots come from their owners."""
        sh = self.sh
        sh["slots"][self.rank] = out[self.rank]
        sh["bar"].wait()
        for r in range(self.world):
            if r != self.rank:
                out[r].copy_(sh["slots"][r])
        torch.cuda.current_stream().synchronize()
        sh["bar"].wait()
        return [None] * self.world


def lora_state(cfg, seed=33, dim=8):
    """Synthetic LoRA state dict in the reference's naming (lora_utils.py:83-145): fused qkv with 3 up-blocks, kv_linear with 2,
    plain up-projections elsewhere; alpha_scale buffers as LoRAModule registers them."""
    g = torch.Generator().manual_seed(seed)
    C, Hd = cfg.hidden_size, cfg.ffn_hidden
    sd = {}
    H = "___lorahyphen___"
    for i in range(cfg.depth):
        for mod, (o, k, nsep) in {f"blocks.{i}.attn.qkv": (3 * C, C, 3), f"blocks.{i}.attn.proj": (C, C, 1),
                                  f"blocks.{i}.cross_attn.kv_linear": (2 * C, C, 2), f"blocks.{i}.ffn.w1": (Hd, C, 1),
                                  f"blocks.{i}.ffn.w2": (C, Hd, 1)}.items():
            name = "lora" + H + mod.replace(".", H)
            sd[name + ".lora_down.weight"] = torch.randn(nsep * dim, k, generator=g) / k ** 0.5
            if nsep > 1:
                for b in range(nsep):
                    sd[name + f".lora_up.blocks.{b}.weight"] = torch.randn(o // nsep, dim, generator=g) * 0.3
            else:
                sd[name + ".lora_up.weight"] = torch.randn(o, dim, generator=g) * 0.3
            sd[name + ".alpha_scale"] = torch.tensor(0.5)
    return sd


# ---- block-sparse attention cases of tests/golden/g18_bsa_triton.npz (tools/make_goldens.py bsa_triton) ---------------------------------
# name: heads, Sq, Sk, block, dtype; inputs are drawn from a seeded CPU generator and rounded to bf16-representable values (fp16 for the
# fp16 case), so that the reference's kernel (fp32 arithmetic in the Triton interpreter) and the HIP kernel (bf16 tensors) see the same numbers
BSA_TRITON_CASES = {
    "k128": dict(H=2, Sq=384, Sk=640, block=128, dtype="f32", sparsity=0.6, seed=101),      # gating + top-k + kernel (flash_attn_bsa)
    "k64": dict(H=2, Sq=256, Sk=384, block=64, dtype="f32", sparsity=0.5, seed=102),        # 64-token blocks (BLOCK_N_LG=64 preset)
    "varlen": dict(H=2, Sq=384, Sk=512, block=128, dtype="f32", sparsity=None, seed=103),   # hand-made variable-length lists incl. empty rows
    "half": dict(H=1, Sq=256, Sk=384, block=128, dtype="f16", sparsity=0.4, seed=104),      # half inputs: p is cast to the value dtype before P V
    "thw": dict(H=2, Sq=512, Sk=512, block=128, dtype="f32", sparsity=0.5, seed=105, grid=(4, 8, 16), chunk=(4, 4, 8)),  # flash_attn_bsa_3d
}


def bsa_triton_inputs(name):
    """-> q [H, Sq, 128], k, v [H, Sk, 128] float32 tensors holding bf16- (fp16-) representable values."""
    import torch
    c = BSA_TRITON_CASES[name]
    g = torch.Generator().manual_seed(c["seed"])
    half = torch.float16 if c["dtype"] == "f16" else torch.bfloat16
    q = torch.randn(c["H"], c["Sq"], 128, generator=g)
    k = torch.randn(c["H"], c["Sk"], 128, generator=g) + 0.5 * torch.randn(c["H"], 1, 128, generator=g)
    v = torch.randn(c["H"], c["Sk"], 128, generator=g)
    return tuple(t.to(half).float() for t in (q, k, v))


def bsa_varlen_lists(name):
    """The hand-made selection of the "varlen" case: sorted-by-nothing index rows + per-row counts (0 = empty selection)."""
    import torch
    c = BSA_TRITON_CASES[name]
    nq, nk = c["Sq"] // c["block"], c["Sk"] // c["block"]
    g = torch.Generator().manual_seed(c["seed"] + 1000)
    idx = torch.stack([torch.stack([torch.randperm(nk, generator=g) for _ in range(nq)]) for _ in range(c["H"])])
    lens = torch.tensor([[0, 1, nk], [2, 0, 3]][:c["H"]], dtype=torch.int32)[:, :nq]
    return idx, lens
