"""Record golden vectors from the UNMODIFIED reference (runs only in the build container, where /root/reference exists).

  python tools/make_goldens.py            # g1, g4, g6, g10 (Wan sampler state machine, FLF metric, harness)
  python tools/make_goldens.py dit | vae   # g7, g8 (in-tree Wan DiT / VAE twins)
  python tools/make_goldens.py vae_akw     # g8b (the executed class: diffusers' AutoencoderKLWan as vendored under longcat_video/modules)
  python tools/make_goldens.py longcat | longcat_pipe | longcat_lora | longcat_refine      # g11, g12, g13, g15 (LongCat DiT, guided
                                           # i2v trajectories, run-time LoRA, refine-pass trajectories)
  TORCHDYNAMO_DISABLE=1 python tools/make_goldens.py bsa | bsa_cdf | bsa_cdf_bf16           # g14, g14b, g14c (block-sparse gating helpers;
                                           # g14c = the cdf counts on bf16 scores, the dtype the reference's model hands them)
  python tools/make_goldens.py warp | warp_cams                                             # g16, g16b (stage-1 forward warp, cameras)
  TRITON_INTERPRET=1 TORCHDYNAMO_DISABLE=1 python tools/make_goldens.py longcat_bsa0       # g11b (LongCat DiT whose self-attention is the
                                           # reference's own Triton kernel at sparsity 0, through the interpreter)
  TRITON_INTERPRET=1 TORCHDYNAMO_DISABLE=1 python tools/make_goldens.py bsa_triton         # g18 (the reference's Triton sparse-attention
                                           # kernel itself, executed on CPU tensors by Triton's interpreter)

The reference is imported through tools/refshim (a stub `diffusers` with base classes only); its DiT / VAE / encoders
are replaced by the deterministic fakes of tests/fakes.py, so the recorded trajectories pin the *sampler state machine*
(scheduler arithmetic, IRR re-noise, FLF gate, DSG, dtype casts, RNG draw order).  Fixtures are data only.
"""
from __future__ import annotations

import ast
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/wan_for_worldforge"
sys.path.insert(0, os.path.join(ROOT, "tools", "refshim"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from tests.fakes import FakeDiT, FakeVAE, synthetic_ref_and_mask  # noqa: E402


def _longcat_paths():
    """LongCat generators only: the reference package + the plain-softmax stand-in for the third-party `flash_attn` package its
    attention module imports.  (Not on sys.path for the Wan generators: the Wan twin's attention.py would take
    FLASH_ATTN_2_AVAILABLE = True from it and route fp32 tensors into flash_attention()'s half-dtype assert, attention.py:53.)"""
    for p in ("/root/reference/longcat_for_worldforge", os.path.join(ROOT, "tools", "refshim_flash")):
        if p not in sys.path:
            sys.path.insert(0, p)


def t2n(t):
    if t.dtype == torch.bfloat16:
        return t.float().numpy()
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------------------
def g_schedules(out_dir=OUT):
    from utils.scheduling_unipc_multistep_clean import UniPCMultistepScheduler

    out = {}
    for n in (4, 16, 50):
        for shift in (3.0, 5.0):
            s = UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=shift)
            s.set_timesteps(n)
            k = f"n{n}_s{int(shift)}"
            out[k + "_timesteps"] = s.timesteps.numpy()
            out[k + "_sigmas"] = s.sigmas.numpy()
            out[k + "_rsig"] = s.resample_sigmas.numpy()
            out[k + "_rts"] = s.resample_timesteps.numpy()
    np.savez_compressed(os.path.join(out_dir, "g1_schedules.npz"), **out)
    print("g1_schedules", len(out))


# ------------------------------------------------------------------------------------------------------------
class _ImgProc:
    def __call__(self, images=None, return_tensors="pt"):
        class _D(dict):
            def to(self, device):
                return self

        return _D(pixel_values=torch.zeros(1, 3, 4, 4))


class _ImgEnc:
    def __init__(self, embeds):
        self.embeds = embeds

    def __call__(self, pixel_values=None, output_hidden_states=True):
        return SimpleNamespace(hidden_states=[None, self.embeds, None])


PIPE_CASES = {
    # name: (steps, R, guide, round, flf, omega, omega_resample, cfg, F, H, W, guided, shift)
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
    ref[:, :, 0] = image  # first frame of the warped sequence is the input image
    pe = torch.randn(1, 16, 32, generator=g).to(torch.bfloat16)
    ne = torch.randn(1, 16, 32, generator=g).to(torch.bfloat16)
    ie = torch.randn(1, 8, 16, generator=g)
    return image, ref, mask, pe, ne, ie


def g_pipeline():
    from utils.pipeline_wan_i2v_clean import WanImageToVideoPipeline
    from utils.scheduling_unipc_multistep_clean import UniPCMultistepScheduler

    for name, c in PIPE_CASES.items():
        image, ref, mask, pe, ne, ie = case_inputs(c)
        dit, vae = FakeDiT(), FakeVAE()
        sch = UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=c["shift"])
        pipe = WanImageToVideoPipeline(tokenizer=None, text_encoder=None, image_encoder=_ImgEnc(ie),
                                       image_processor=_ImgProc(), transformer=dit, vae=vae, scheduler=sch)
        rec = {}
        calls = []
        orig_step = sch.step

        def wrapped(*a, **k):
            o = orig_step(*a, **k)
            calls.append((t2n(o.prev_sample), t2n(o.pred_x0), str(o.prev_sample.dtype), str(o.pred_x0.dtype)))
            return o

        sch.step = wrapped
        lat_steps = []

        def cb(p, i, t, kw):
            lat_steps.append((t2n(kw["latents"]), str(kw["latents"].dtype)))
            return {}

        gen = torch.manual_seed(42)
        out = pipe(image=image, prompt=None, negative_prompt=None, height=c["H"], width=c["W"], num_frames=c["F"],
                   num_inference_steps=c["steps"], guidance_scale=c["cfg"], generator=gen, prompt_embeds=pe,
                   negative_prompt_embeds=ne, output_type="np", video_ref=ref, mask=mask, guided=c["guided"],
                   resample_steps=c["R"], guide_steps=c["guide"], omega=c["omega"], omega_resample=c["omega_r"],
                   resample_round=c["rnd"], use_pca_channel_selection=c["flf"], static=True, callback_on_step_end=cb)
        frames = out.frames
        rec["frames"] = np.asarray(frames, dtype=np.float32)
        rec["n_calls"] = np.array([dit.calls, vae.n_enc, vae.n_dec])
        for j, (p, x0, dp, dx) in enumerate(calls):
            rec[f"call{j}_prev"] = p
            rec[f"call{j}_x0"] = x0
            rec[f"call{j}_dtypes"] = np.array([dp, dx])
        for j, (l, d) in enumerate(lat_steps):
            rec[f"lat{j}"] = l
            rec[f"lat{j}_dtype"] = np.array([d])
        rec["n_step_calls"] = np.array([len(calls)])
        np.savez_compressed(os.path.join(OUT, f"g6_pipe_{name}.npz"), **rec)
        print("g6", name, "dit/enc/dec calls", rec["n_calls"], "step calls", len(calls), "frames", rec["frames"].shape)


# ------------------------------------------------------------------------------------------------------------
def g_flf():
    from utils.scheduling_unipc_multistep_clean import VideoMotionPCASelector

    sel = VideoMotionPCASelector()
    g = torch.Generator().manual_seed(7)
    out = {}
    # flow metric on random 2-component and 1-component flows of several magnitudes
    for k, (scale_r, scale_c, cr, cc) in enumerate([(1.0, 1.0, 2, 2), (6.0, 5.0, 2, 2), (0.2, 0.3, 1, 1), (4.0, 0.5, 2, 1),
                                                    (12.0, 12.0, 1, 1)]):
        r = torch.randn(1, 6, cr, 10, 12, generator=g) * scale_r
        c = r * 0.7 + torch.randn(1, 6, cc, 10, 12, generator=g) * scale_c if cr == cc else torch.randn(
            1, 6, cc, 10, 12, generator=g) * scale_c
        out[f"fm{k}_ref"] = r.numpy()
        out[f"fm{k}_chan"] = c.numpy()
        out[f"fm{k}_sim"] = np.array([sel._compute_flow_metrics(r, c, None)])
    # threshold logic with injected similarities
    sims_sets = [torch.rand(16, generator=g).numpy().astype(np.float64) for _ in range(3)]
    sims_sets.append(np.full(16, 0.5))
    sims_sets.append(np.concatenate([np.full(8, 0.1), np.full(8, 0.9)]))
    for si, sims in enumerate(sims_sets):
        out[f"sel{si}_sims"] = sims
        for step in (0, 1, 2, 3, 5, 6, 10, 11, 30):
            sel2 = VideoMotionPCASelector()
            sel2._compute_channel_correlations = lambda *a, _s=sims, **k: list(_s)
            x = torch.zeros(1, 16, 3, 4, 4)
            ch = sel2.select_motion_related_channels(x, x, None, current_step=step, use_optical_flow=False)
            out[f"sel{si}_step{step}"] = np.array(ch, dtype=np.int64)
    # end-to-end temporal-difference branch (what the reference executes without cv2)
    pred = torch.randn(1, 16, 5, 6, 8, generator=g)
    enc = pred * 0.8 + 0.3 * torch.randn(1, 16, 5, 6, 8, generator=g)
    enc[:, 3] = torch.randn(5, 6, 8, generator=g) * 9
    enc[:, 9] = torch.randn(5, 6, 8, generator=g) * 7
    for step in (3, 8, 12):
        ch = VideoMotionPCASelector().select_motion_related_channels(pred, enc, None, current_step=step,
                                                                     use_optical_flow=True)
        out[f"e2e_step{step}"] = np.array(ch, dtype=np.int64)
    out["e2e_pred"] = pred.numpy()
    out["e2e_enc"] = enc.numpy()
    sims = VideoMotionPCASelector()._compute_channel_correlations(
        pred, enc, None, True,
        [(enc[:, c:c + 1, 1:] - enc[:, c:c + 1, :-1]).permute(0, 2, 1, 3, 4) for c in range(16)])
    out["e2e_sims"] = np.array(sims)
    np.savez_compressed(os.path.join(OUT, "g4_flf.npz"), **out)
    print("g4_flf", len(out))


# ------------------------------------------------------------------------------------------------------------
def _extract_functions(path, names):
    """Execute only the named top-level function definitions of a reference script (INFER runs its CLI at import)."""
    src = open(path).read()
    tree = ast.parse(src)
    ns = {}
    exec("import os, glob\nimport numpy as np\nfrom PIL import Image\nfrom scipy.ndimage import distance_transform_edt\n", ns)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns


def g_harness():
    ns = _extract_functions(os.path.join(REF, "infer_worldforge.py"), {"soften_mask", "read_frames_from_directory"})
    soften = ns["soften_mask"]
    out = {}
    yy, xx = np.mgrid[0:48, 0:64]
    disc = ((yy - 24) ** 2 + (xx - 30) ** 2 < 18 ** 2).astype(np.float64)
    half = (xx < 37).astype(np.float64)
    grey = half.copy()
    grey[:, 30:37] = 0.4  # non-binary edge values as produced by PIL bicubic resize (any non-zero counts as True)
    masks = np.stack([disc, half, np.ones_like(disc), np.zeros_like(disc), grey])
    out["masks"] = masks
    for decay in ("linear", "exponential", "sine", "cosine"):
        for d in (5, 15):
            out[f"soft_{decay}_{d}"] = soften(masks, d, decay)
    # size rule INFER:218-222
    rows = []
    for (ih, iw) in [(720, 1280), (480, 832), (512, 960), (1080, 1920), (1000, 1000), (832, 480)]:
        for max_area in (480 * 832, 720 * 1280):
            ar = ih / iw
            h = round(np.sqrt(max_area * ar)) // 16 * 16
            w = round(np.sqrt(max_area / ar)) // 16 * 16
            rows.append([ih, iw, max_area, h, w])
    out["size_rule"] = np.array(rows)
    np.savez_compressed(os.path.join(OUT, "g10_harness.npz"), **out)
    print("g10_harness", len(out))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["sched", "pipe", "flf", "harness"]
    if "sched" in which:
        g_schedules()
    if "pipe" in which:
        g_pipeline()
    if "flf" in which:
        g_flf()
    if "harness" in which:
        g_harness()


def _load_wan_module(name):
    """Import /root/reference/.../wan/modules/<name>.py without running wan/__init__.py (which needs easydict etc.)."""
    import importlib
    import types

    for pkg, path in (("wan", os.path.join(REF, "wan")), ("wan.modules", os.path.join(REF, "wan", "modules"))):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [path]
            sys.modules[pkg] = m
    return importlib.import_module(f"wan.modules.{name}")


# ------------------------------------------------------------------------------------------------------------
def g_dit(out_dir=OUT):
    """G7: tiny WanModel (in-tree twin, fp32, SDPA fallback for flash_attention) with the oracle's synthetic weights."""
    wattn = _load_wan_module("attention")
    wmodel = _load_wan_module("model")
    from oracle import dit as odit

    wmodel.flash_attention = lambda q, k, v, k_lens=None, window_size=(-1, -1): wattn.attention(
        q, k, v, k_lens=k_lens, window_size=window_size, fa_version=None)
    # attention.py:133-179 casts to `dtype` (bf16) by default before SDPA; keep fp32 so the golden is a pure fp32 statement
    orig_attention = wattn.attention
    wattn_attention = lambda q, k, v, **kw: orig_attention(q, k, v, **{**kw, "dtype": torch.float32})
    wmodel.flash_attention = lambda q, k, v, k_lens=None, window_size=(-1, -1): wattn_attention(
        q, k, v, k_lens=k_lens, window_size=window_size, fa_version=None)
    out = {}
    for name, (dim, heads, ffn, layers, T, h, w) in {"tiny": (256, 2, 512, 2, 2, 8, 12), "odd": (384, 3, 640, 1, 3, 6, 10)}.items():
        cfg = odit.DiTConfig(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64, img_dim=1280)
        W = odit.random_weights(cfg, seed=11)
        m = wmodel.WanModel(model_type="i2v", in_dim=36, dim=dim, ffn_dim=ffn, freq_dim=256, text_dim=64, out_dim=16,
                            num_heads=heads, num_layers=layers)
        missing, unexpected = m.load_state_dict(W, strict=True), None
        m.eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(16, T, h, w, generator=g)
        y = torch.randn(20, T, h, w, generator=g)
        ctx = torch.randn(40, 64, generator=g)
        clip = torch.randn(1, 257, 1280, generator=g)
        t = torch.tensor([749])
        with torch.no_grad():
            o = m([x], t, [ctx], seq_len=T * (h // 2) * (w // 2), clip_fea=clip, y=[y])[0]
        out[f"{name}_x"] = torch.cat([x, y]).numpy()
        out[f"{name}_ctx"] = ctx.numpy()
        out[f"{name}_clip"] = clip[0].numpy()
        out[f"{name}_out"] = o.numpy()
        out[f"{name}_cfg"] = np.array([dim, heads, ffn, layers, T, h, w])
    np.savez_compressed(os.path.join(out_dir, "g7_dit.npz"), **out)
    print("g7_dit", {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__" and "dit" in sys.argv[1:]:
    g_dit()


# ------------------------------------------------------------------------------------------------------------
def g_vae():
    """G8: the in-tree WanVAE_ (real config: dim 96, z 16, 127 M parameters, chunked + feat_cache) with the oracle's
    synthetic weights; small videos so the fixture stays small.  Also pins the state_dict names / shapes."""
    wvae = _load_wan_module("vae")
    from oracle import vae as ovae

    m = wvae.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                     temperal_downsample=[False, True, True], dropout=0.0)
    W = ovae.random_weights(seed=5)
    sd = m.state_dict()
    assert set(sd) == set(W), (set(sd) ^ set(W))
    assert all(tuple(sd[k].shape) == tuple(W[k].shape) for k in sd)
    m.load_state_dict(W, strict=True)
    m.eval()
    out = {}
    g = torch.Generator().manual_seed(9)
    cases = {"f9_32x32": (9, 32, 32), "f5_48x40": (5, 48, 40), "f1_32x32": (1, 32, 32), "f17_16x24": (17, 16, 24)}
    with torch.no_grad():
        for name, (Fr, H, Wd) in cases.items():
            x = torch.rand(1, 3, Fr, H, Wd, generator=g) * 2 - 1
            mu = m.encode(x, [0.0, 1.0])
            T = (Fr - 1) // 4 + 1
            z = torch.randn(1, 16, T, H // 8, Wd // 8, generator=g)
            dec = m.decode(z, [0.0, 1.0]).clamp(-1, 1)
            out[f"{name}_x"], out[f"{name}_mu"] = x.numpy(), mu.numpy()
            out[f"{name}_z"], out[f"{name}_dec"] = z.numpy(), dec.numpy()
            print("g8", name, tuple(mu.shape), tuple(dec.shape), float(mu.abs().max()), float(dec.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "g8_vae.npz"), **out)


if __name__ == "__main__" and "vae" in sys.argv[1:]:
    g_vae()


def _load_akw():
    """diffusers' AutoencoderKLWan as vendored at longcat_video/modules/autoencoder_kl_wan.py (imports diffusers base classes only ->
    tools/refshim) -- the class the Wan path executes (INFER:185-189) and LongCat loads."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "akw_ref", "/root/reference/longcat_for_worldforge/longcat_video/modules/autoencoder_kl_wan.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def g_vae_akw(out_dir=OUT):
    """G8b: the executed class.  The oracle's synthetic weights (twin names) are loaded into the unmodified diffusers-layout class
    through the inverse of worldforge_amd.vae.diffusers_key_map; encode(x).latent_dist.mode() / decode(z).sample (clamped, :1222;
    chunked _encode :1145-1170) on the g8 inputs.  Also records the class's parameter names + shapes (the key-map contract)."""
    akw = _load_akw()
    from oracle import vae as ovae
    from worldforge_amd.vae import diffusers_key_map

    m = akw.AutoencoderKLWan()
    m.eval()
    sd = m.state_dict()
    W = ovae.random_weights(seed=5)
    inv = {v: k for k, v in diffusers_key_map().items()}
    new = {}
    for k, v in W.items():
        base, _, leaf = k.rpartition(".")
        dk = f"{inv[base]}.{leaf}"
        new[dk] = v.reshape(sd[dk].shape)
    assert set(new) == set(sd), set(new) ^ set(sd)
    m.load_state_dict(new, strict=True)
    g8 = np.load(os.path.join(OUT, "g8_vae.npz"))
    out = {"param_names": np.array(sorted(sd)), "param_shapes": np.array([",".join(map(str, sd[k].shape)) for k in sorted(sd)])}
    with torch.no_grad():
        for name in ("f9_32x32", "f5_48x40", "f1_32x32", "f17_16x24"):
            x, z = torch.from_numpy(g8[f"{name}_x"]), torch.from_numpy(g8[f"{name}_z"])
            mu = m.encode(x).latent_dist.mode()
            dec = m.decode(z, return_dict=False)[0]
            out[f"{name}_mu"], out[f"{name}_dec"] = mu.numpy(), dec.numpy()
            print("g8b", name, float((mu - torch.from_numpy(g8[f"{name}_mu"])).abs().max()),
                  float((dec - torch.from_numpy(g8[f"{name}_dec"])).abs().max()))
    np.savez_compressed(os.path.join(out_dir, "g8b_vae_akw.npz"), **out)


if __name__ == "__main__" and "vae_akw" in sys.argv[1:]:
    g_vae_akw()


def g_vae_akw_bf16(out_dir=OUT):
    """G8c: the executed class IN THE DTYPE THE LONGCAT ENTRY LOADS IT (run_longcat_worldforge_single.py:205
    `AutoencoderKLWan.from_pretrained(..., torch_dtype=torch.bfloat16)`): the g8b model `.to(torch.bfloat16)`, inputs cast with the
    `.to(dtype=vae.dtype)` of the LongCat fuse_latents (scheduling_flow_match_euler_discrete.py:1124, 1166), run by eager PyTorch on
    CPU.  Weights, activations, the residual stream and the accumulator hand-offs between layers are all bf16 there.  The fixture
    holds the outputs (as f32) and their distance from the class's own fp32 run (g8b): the yardstick a reduced-precision VAE mode of
    the build is held against (tests/test_gpu_vae.py: no farther from fp32 than the reference's own arithmetic is)."""
    akw = _load_akw()
    from oracle import vae as ovae
    from worldforge_amd.vae import diffusers_key_map

    m = akw.AutoencoderKLWan()
    m.eval()
    sd = m.state_dict()
    W = ovae.random_weights(seed=5)
    inv = {v: k for k, v in diffusers_key_map().items()}
    m.load_state_dict({f"{inv[k.rpartition('.')[0]]}.{k.rpartition('.')[2]}": v.reshape(sd[f"{inv[k.rpartition('.')[0]]}.{k.rpartition('.')[2]}"].shape)
                       for k, v in W.items()}, strict=True)
    m = m.to(torch.bfloat16)
    g8 = np.load(os.path.join(OUT, "g8_vae.npz"))
    g8b = np.load(os.path.join(OUT, "g8b_vae_akw.npz"))
    out = {}
    with torch.no_grad():
        for name in ("f9_32x32", "f5_48x40", "f1_32x32", "f17_16x24"):
            x, z = torch.from_numpy(g8[f"{name}_x"]), torch.from_numpy(g8[f"{name}_z"])
            mu = m.encode(x.to(dtype=m.dtype)).latent_dist.mode()
            dec = m.decode(z.to(dtype=m.dtype), return_dict=False)[0]
            assert mu.dtype == torch.bfloat16 and dec.dtype == torch.bfloat16
            out[f"{name}_mu"], out[f"{name}_dec"] = t2n(mu), t2n(dec)
            mu32, dec32 = torch.from_numpy(g8b[f"{name}_mu"]), torch.from_numpy(g8b[f"{name}_dec"])
            out[f"{name}_mu_rel"] = np.float64(((mu.float() - mu32).norm() / mu32.norm()).item())
            out[f"{name}_dec_rel"] = np.float64(((dec.float() - dec32).norm() / dec32.norm()).item())
            print("g8c", name, "rel L2 of the bf16 run from the fp32 run: mu", out[f"{name}_mu_rel"], "dec", out[f"{name}_dec_rel"])
    np.savez_compressed(os.path.join(out_dir, "g8c_vae_akw_bf16.npz"), **out)


if __name__ == "__main__" and "vae_akw_bf16" in sys.argv[1:]:
    g_vae_akw_bf16()


# ------------------------------------------------------------------------------------------------------------
def g_longcat_dit():
    """G11: the unmodified LongCatVideoTransformer3DModel (fp32, CPU; its flash-attn calls served by tools/refshim_flash/flash_attn) with
    the oracle's synthetic weights, called the way generate_i2v calls it (pipeline_longcat_video.py:857-873: cond + uncond batch,
    per-frame timesteps with frame 0 at t = 0, num_cond_latents = 1, caption masks), plus one call without condition frames."""
    import warnings

    _longcat_paths()
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from oracle import longcat_dit as olc

    out = {}
    cases = {"tiny": dict(C=256, heads=2, depth=2, cap=64, ct=64, T=3, h=8, w=12, ncond=1, zero_pad=False),
             "odd": dict(C=384, heads=3, depth=1, cap=96, ct=128, T=2, h=6, w=10, ncond=0, zero_pad=False),
             "zpad": dict(C=256, heads=2, depth=1, cap=64, ct=64, T=2, h=4, w=8, ncond=1, zero_pad=True)}
    for name, c in cases.items():
        cfg = olc.LongCatConfig(hidden_size=c["C"], depth=c["depth"], num_heads=c["heads"], caption_channels=c["cap"],
                                adaln_tembed_dim=c["ct"], text_tokens_zero_pad=c["zero_pad"])
        W = olc.random_weights(cfg, seed=21)
        m = LongCatVideoTransformer3DModel(hidden_size=c["C"], depth=c["depth"], num_heads=c["heads"], caption_channels=c["cap"],
                                           adaln_tembed_dim=c["ct"], enable_flashattn2=True, cp_split_hw=[1, 1],
                                           text_tokens_zero_pad=c["zero_pad"])
        sd = m.state_dict()
        assert set(sd) == set(W), (set(sd) ^ set(W))
        assert all(tuple(sd[k].shape) == tuple(W[k].shape) for k in sd)
        m.load_state_dict(W, strict=True)
        m.eval()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, 16, c["T"], c["h"], c["w"], generator=g)
        cap = torch.randn(2, 1, 24, c["cap"], generator=g)
        mask = torch.zeros(2, 24, dtype=torch.int64)
        mask[0, :9] = 1   # negative prompt: 9 valid tokens
        mask[1, :17] = 1  # positive prompt: 17 valid tokens
        ts = torch.full((2, c["T"]), 637.0)
        if c["ncond"]:
            ts[:, :1] = 0
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = m(hidden_states=torch.cat([x, x]), timestep=ts, encoder_hidden_states=cap, encoder_attention_mask=mask,
                  num_cond_latents=c["ncond"])
        out[f"{name}_x"], out[f"{name}_cap"], out[f"{name}_mask"] = x[0].numpy(), cap[:, 0].numpy(), mask.numpy()
        out[f"{name}_ts"], out[f"{name}_out"] = ts.numpy(), o.numpy()
        out[f"{name}_cfg"] = np.array([c["C"], c["heads"], c["depth"], c["cap"], c["ct"], c["ncond"], int(c["zero_pad"])])
        print("g11", name, tuple(o.shape), float(o.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "g11_longcat_dit.npz"), **out)


if __name__ == "__main__" and "longcat" in sys.argv[1:]:
    g_longcat_dit()


def g_longcat_dit_bsa0(out_dir=OUT):
    """G11b: the unmodified LongCatVideoTransformer3DModel with `enable_bsa=True, sparsity = 0` -- every key block
    selected, so its self-attention is a DENSE softmax computed by the REFERENCE'S OWN Triton kernel (flash_attn_bsa_3d ->
    _attn_fwd_bsa_varlen_align, bsa_interface.py:612-659 / flash_attn_bsa_varlen_mask.py:174-285) executed on CPU tensors by Triton's
    interpreter, instead of the builder's plain-softmax stand-in for flash-attn that served g11.  (The cross-attention has no such path in
    the reference -- attention.py:218-246 calls flash_attn_varlen_func or xformers only -- so it stays on the stand-in: 24 caption keys.)
    fp32, 4 x 8 x 8 = 256 tokens = four 64-token blocks (chunk 4 x 4 x 4).  Needs TRITON_INTERPRET=1 TORCHDYNAMO_DISABLE=1."""
    assert os.environ.get("TRITON_INTERPRET") == "1", "run with TRITON_INTERPRET=1 TORCHDYNAMO_DISABLE=1"
    import warnings

    _longcat_paths()
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from oracle import longcat_dit as olc

    c = dict(C=256, heads=2, depth=2, cap=64, ct=64, T=4, h=16, w=16)
    cfg = olc.LongCatConfig(hidden_size=c["C"], depth=c["depth"], num_heads=c["heads"], caption_channels=c["cap"], adaln_tembed_dim=c["ct"])
    W = olc.random_weights(cfg, seed=21)
    bsa_params = dict(sparsity=0.0, chunk_3d_shape_q=[4, 4, 4], chunk_3d_shape_k=[4, 4, 4])
    m = LongCatVideoTransformer3DModel(hidden_size=c["C"], depth=c["depth"], num_heads=c["heads"], caption_channels=c["cap"],
                                       adaln_tembed_dim=c["ct"], enable_flashattn2=True, enable_bsa=True, bsa_params=bsa_params,
                                       cp_split_hw=[1, 1])
    m.load_state_dict(W, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, c["T"], c["h"], c["w"], generator=g)
    cap = torch.randn(1, 1, 24, c["cap"], generator=g)
    mask = torch.zeros(1, 24, dtype=torch.int64)
    mask[0, :17] = 1
    ts = torch.full((1, c["T"]), 637.0)
    calls = []
    from longcat_video.block_sparse_attention import bsa_interface as B
    orig = B.attn_fwd_bsa_varlen_triton

    def counted(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    B.attn_fwd_bsa_varlen_triton = counted
    try:
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = m(hidden_states=x, timestep=ts, encoder_hidden_states=cap, encoder_attention_mask=mask, num_cond_latents=0)
    finally:
        B.attn_fwd_bsa_varlen_triton = orig
    assert len(calls) == c["depth"], calls      # every block's self-attention went through the reference's Triton kernel
    out = {"x": x[0].numpy(), "cap": cap[:, 0].numpy(), "mask": mask.numpy(), "ts": ts.numpy(), "out": o.numpy(),
           "cfg": np.array([c["C"], c["heads"], c["depth"], c["cap"], c["ct"]]), "triton_calls": np.array(len(calls))}
    np.savez_compressed(os.path.join(out_dir, "g11b_longcat_dit_bsa0.npz"), **out)
    print("g11b", tuple(o.shape), float(o.abs().mean()), "triton self-attention launches:", len(calls))


if __name__ == "__main__" and "longcat_bsa0" in sys.argv[1:]:
    g_longcat_dit_bsa0()


# ------------------------------------------------------------------------------------------------------------
LC_PIPE_CASES = {
    "irr_flf": dict(steps=9, R=2, guide=8, rnd=8, flf=True, omega=1.8, omega_r=1.0, cfg=4.0, F=9, H=32, W=32, guided=True, shift=1.0,
                    distill=False, maxrep=None),
    "dsg_shift": dict(steps=6, R=3, guide=3, rnd=5, flf=False, omega=2.5, omega_r=1.5, cfg=3.0, F=5, H=32, W=48, guided=True, shift=8.0,
                      distill=False, maxrep=None),
    "plain": dict(steps=5, R=3, guide=0, rnd=0, flf=False, omega=1.8, omega_r=1.0, cfg=4.0, F=5, H=32, W=32, guided=False, shift=3.0,
                  distill=False, maxrep=None),
    "nocfg_distill": dict(steps=6, R=2, guide=6, rnd=6, flf=True, omega=1.8, omega_r=1.0, cfg=1.0, F=9, H=32, W=32, guided=True,
                          shift=1.0, distill=True, maxrep=2),
}


def lc_case_inputs(c):
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


def g_longcat_pipe(vae_dtype=torch.float32, cases=None, fmt="g12_longcat_pipe_{}.npz"):
    """G12: LongCatVideoPipeline.generate_i2v + FlowMatchEulerDiscreteScheduler, unmodified, with deterministic stand-ins for the DiT,
    the VAE and the text encoder (encode_prompt) and a fixed target size instead of the resolution-bucket lookup.
    G12b (`longcat_pipe_bf16vae`): the same with the VAE stand-in IN BF16, the dtype the LongCat entry loads its VAE in
    (run_longcat_worldforge_single.py:205): pins the `.to(dtype=vae.dtype)` hand-offs of fuse_latents
    (scheduling_flow_match_euler_discrete.py:1124, 1166), the bf16 pixel blend (:1152-1164), the bf16 posterior sample of
    prepare_latents and the final `latents.to(self.vae.dtype)` + bf16 de-normalisation (pipeline_longcat_video.py:999-1001)."""
    from tests.fakes import FakeLongCatDiT

    _longcat_paths()
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline

    for name, c in LC_PIPE_CASES.items():
        if cases is not None and name not in cases:
            continue
        image, ref, mask, pe, pm, ne, nm = lc_case_inputs(c)
        dit, vae = FakeLongCatDiT(), FakeVAE(vae_dtype)
        sch = FlowMatchEulerDiscreteScheduler(shift=c["shift"])
        pipe = LongCatVideoPipeline(tokenizer=None, text_encoder=None, vae=vae, scheduler=sch, dit=dit)
        pipe.device = "cpu"
        pipe.encode_prompt = lambda **kw: (pe, pm, ne if kw["do_classifier_free_guidance"] else None,
                                           nm if kw["do_classifier_free_guidance"] else None)
        pipe.get_condition_shape = lambda *a, **k: (c["H"], c["W"])
        rec, calls = {}, []
        orig_step = sch.step

        def wrapped(*a, **k):
            o = orig_step(*a, **k)
            calls.append((t2n(o.prev_sample), t2n(o.pred_x0)))
            return o

        sch.step = wrapped
        orig_prep = pipe.prepare_latents

        def prep(*a, **k):
            lat = orig_prep(*a, **k)
            rec["latents0"] = t2n(lat).copy()
            return lat

        pipe.prepare_latents = prep
        gen = torch.manual_seed(42)
        frames = pipe.generate_i2v(image=image, prompt="p", negative_prompt="n", resolution="480p", num_frames=c["F"],
                                   num_inference_steps=c["steps"], use_distill=c["distill"], guidance_scale=c["cfg"], generator=gen,
                                   output_type="np", video_ref=ref, mask=mask, guided=c["guided"], resample_steps=c["R"],
                                   guide_steps=c["guide"], resample_round=c["rnd"], omega=c["omega"], omega_resample=c["omega_r"],
                                   use_pca_channel_selection=c["flf"], static=True, max_replace_threshold=c["maxrep"])
        rec["frames"] = np.asarray(frames, dtype=np.float32)
        rec["n_calls"] = np.array([dit.calls, vae.n_enc, vae.n_dec])
        rec["sigmas"], rec["timesteps"] = sch.sigmas.numpy(), sch.timesteps.numpy()
        for j, (p, x0) in enumerate(calls):
            rec[f"call{j}_prev"], rec[f"call{j}_x0"] = p, x0
        rec["n_step_calls"] = np.array([len(calls)])
        np.savez_compressed(os.path.join(OUT, fmt.format(name)), **rec)
        print(fmt.format(name), name, "dit/enc/dec calls", rec["n_calls"], "step calls", len(calls), "frames", rec["frames"].shape)


if __name__ == "__main__" and "longcat_pipe" in sys.argv[1:]:
    g_longcat_pipe()
if __name__ == "__main__" and "longcat_pipe_bf16vae" in sys.argv[1:]:
    g_longcat_pipe(torch.bfloat16, cases=("irr_flf", "nocfg_distill"), fmt="g12b_longcat_pipe_{}_vaebf16.npz")


# ------------------------------------------------------------------------------------------------------------
def g_longcat_lora():
    """G13: the reference DiT with a LoRA network enabled the reference's way (create_lora_network + enable_loras, LCD:189-247)."""
    import warnings

    _longcat_paths()
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.lora_utils import create_lora_network
    from oracle import longcat_dit as olc
    from tests.fakes import lora_state

    cfg = olc.LongCatConfig(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(cfg, seed=21)
    m = LongCatVideoTransformer3DModel(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64,
                                       enable_flashattn2=True, cp_split_hw=[1, 1])
    m.load_state_dict(W, strict=True)
    m.eval()
    lsd = lora_state(cfg)
    net = create_lora_network(transformer=m, lora_network_state_dict_loaded=lsd, multiplier=0.8, network_dim=8, network_alpha=4)
    net.load_state_dict(lsd, strict=True)
    m.lora_dict["k"] = net
    m.enable_loras(["k"])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 3, 8, 12, generator=g)
    cap = torch.randn(1, 1, 24, 64, generator=g)
    mask = torch.zeros(1, 24, dtype=torch.int64)
    mask[0, :17] = 1
    ts = torch.full((1, 3), 500.0)
    ts[:, :1] = 0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = m(hidden_states=x, timestep=ts, encoder_hidden_states=cap, encoder_attention_mask=mask, num_cond_latents=1)
        m.disable_all_loras()
        o0 = m(hidden_states=x, timestep=ts, encoder_hidden_states=cap, encoder_attention_mask=mask, num_cond_latents=1)
    np.savez_compressed(os.path.join(OUT, "g13_longcat_lora.npz"), x=x[0].numpy(), cap=cap[0, 0].numpy(), mask=mask[0].numpy(),
                        ts=ts[0].numpy(), out=o[0].numpy(), out_base=o0[0].numpy())
    print("g13", tuple(o.shape), float((o - o0).abs().mean()), float(o0.abs().mean()))


if __name__ == "__main__" and "longcat_lora" in sys.argv[1:]:
    g_longcat_lora()


# ------------------------------------------------------------------------------------------------------------
def g_bsa():
    """G14: the reference's own BSA helper functions (permutes, mean pooling, top-k block selection; eager: TORCHDYNAMO_DISABLE=1)."""
    _longcat_paths()
    from longcat_video.block_sparse_attention import bsa_interface as B

    g = torch.Generator().manual_seed(9)
    out = {}
    for name, (T, H, W, t, h, w) in {"a": (8, 8, 16, 4, 4, 8), "b": (4, 12, 8, 2, 4, 4)}.items():
        x = torch.arange(T * H * W, dtype=torch.float32).view(1, 1, -1, 1)
        xb = B.rearrange_THW_to_3d_block(x, T // t, H // h, W // w, t, h, w, 1)
        out[f"{name}_perm"] = xb.flatten().long().numpy()
        assert torch.equal(B.rearrange_3d_block_to_THW(xb, T // t, H // h, W // w, t, h, w, 1), x)
        out[f"{name}_shape"] = np.array([T, H, W, t, h, w])
    for name, (Hh, Sq, Sk, blk, sp, dt) in {"f32": (3, 1024, 1536, 128, 0.75, torch.float32),
                                            "bf16": (2, 768, 768, 128, 0.5, torch.bfloat16)}.items():
        q = torch.randn(1, Hh, Sq, 128, generator=g).to(dt)
        k = (torch.randn(1, Hh, Sk, 128, generator=g) + 0.3 * torch.randn(1, Hh, 1, 128, generator=g)).to(dt)
        qc, kc = B.mean_pooling_compression(q, blk), B.mean_pooling_compression(k, blk)
        idx, lens = B.get_select_indices(qc, kc, sp, None)
        out[f"{name}_q"], out[f"{name}_k"] = q[0].float().numpy(), k[0].float().numpy()
        out[f"{name}_qc"], out[f"{name}_kc"] = qc[0].float().numpy(), kc[0].float().numpy()
        out[f"{name}_idx"], out[f"{name}_lens"] = idx[0].numpy(), lens[0].numpy()
        out[f"{name}_cfg"] = np.array([Hh, Sq, Sk, blk, int(sp * 1000)])
    np.savez_compressed(os.path.join(OUT, "g14_bsa.npz"), **out)
    print("g14", {k: v.shape for k, v in out.items() if k.endswith("_idx")})


if __name__ == "__main__" and "bsa" in sys.argv[1:]:
    g_bsa()


# ------------------------------------------------------------------------------------------------------------
LC_REFINE_CASES = {"spatial": dict(F0=5, H0=32, W0=48, H=64, W=128, steps=8, shift=1.0, t=0.5, sro=True),
                   "spatiotemporal": dict(F0=3, H0=24, W0=40, H=64, W=64, steps=6, shift=4.0, t=0.6, sro=False)}


def lc_refine_inputs(c):
    g = torch.Generator().manual_seed(23)
    frames = (torch.rand(c["F0"], c["H0"], c["W0"], 3, generator=g) * 255).to(torch.uint8)
    image = torch.rand(3, c["H"], c["W"], generator=g)
    pe = torch.randn(1, 1, 12, 32, generator=g).to(torch.bfloat16)
    pm = torch.zeros(1, 12, dtype=torch.int64)
    pm[:, :9] = 1
    return frames, image, pe, pm


def g_longcat_refine(vae_dtype=torch.float32, cases=None, fmt="g15_longcat_refine_{}.npz"):
    """G15: LongCatVideoPipeline.generate_refine, unmodified, with the deterministic DiT / VAE / text-encoder stand-ins.
    G15b (`longcat_refine_bf16vae`): with the VAE stand-in in bf16 (see G12b): the bf16 posterior sample, bf16 normalisation, bf16 noise
    draw and bf16 noise mix of pipeline_longcat_video.py:1430-1433."""
    from tests.fakes import FakeLongCatDiT

    _longcat_paths()
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    import longcat_video.pipeline_longcat_video as lcp
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline

    lcp.torch_gc = lambda: None  # memory housekeeping that calls torch.cuda.ipc_collect() unconditionally (no GPU here); no arithmetic

    for name, c in LC_REFINE_CASES.items():
        if cases is not None and name not in cases:
            continue
        frames, image, pe, pm = lc_refine_inputs(c)
        dit, vae = FakeLongCatDiT(), FakeVAE(vae_dtype)
        sch = FlowMatchEulerDiscreteScheduler(shift=c["shift"])
        pipe = LongCatVideoPipeline(tokenizer=None, text_encoder=None, vae=vae, scheduler=sch, dit=dit)
        pipe.device = "cpu"
        pipe.encode_prompt = lambda **kw: (pe, pm, None, None)
        pipe.get_condition_shape = lambda *a, **k: (c["H"], c["W"])
        rec, lats = {}, []
        orig_step = sch.step

        def wrapped(*a, **k):
            o = orig_step(*a, **k)
            lats.append(t2n(o[0]))
            return o

        sch.step = wrapped
        orig_enc, ups = vae.encode, []

        def enc(x):
            ups.append(t2n(x))
            return orig_enc(x)

        vae.encode = enc
        out = pipe.generate_refine(image=image, prompt="p", stage1_video=[f.numpy() for f in frames], num_cond_frames=1,
                                   num_inference_steps=c["steps"], generator=torch.manual_seed(42), output_type="np", t_thresh=c["t"],
                                   spatial_refine_only=c["sro"])
        rec["frames"] = np.asarray(out, dtype=np.float32)
        rec["video_up"] = ups[0]
        rec["timesteps"], rec["sigmas"] = sch.timesteps.numpy(), sch.sigmas.numpy()
        for j, l in enumerate(lats):
            rec[f"step{j}"] = l
        rec["n"] = np.array([len(lats), dit.calls, vae.n_enc, vae.n_dec])
        np.savez_compressed(os.path.join(OUT, fmt.format(name)), **rec)
        print(fmt.format(name), name, rec["n"], rec["frames"].shape, rec["video_up"].shape)


if __name__ == "__main__" and "longcat_refine" in sys.argv[1:]:
    g_longcat_refine()
if __name__ == "__main__" and "longcat_refine_bf16vae" in sys.argv[1:]:
    g_longcat_refine(torch.bfloat16, cases=("spatial",), fmt="g15b_longcat_refine_{}_vaebf16.npz")


def g_bsa_cdf():
    """G14b: the reference's cdf / cdf+top-k block selection functions (bsa_interface.py:226-263), eager."""
    _longcat_paths()
    from longcat_video.block_sparse_attention import bsa_interface as B

    g = torch.Generator().manual_seed(19)
    out = {}
    for name, (Hh, nq, nk, thr, sp) in {"cdf": (2, 6, 12, 0.6, None), "cdf_topk": (3, 5, 16, 0.3, 0.75)}.items():
        qc = torch.randn(1, Hh, nq, 128, generator=g) * 2.0
        kc = torch.randn(1, Hh, nk, 128, generator=g) * 2.0
        idx, lens = B.get_select_indices(qc, kc, sp, thr)
        out[f"{name}_qc"], out[f"{name}_kc"] = qc[0].numpy(), kc[0].numpy()
        out[f"{name}_idx"], out[f"{name}_lens"] = idx[0].numpy(), lens[0].numpy()
        out[f"{name}_cfg"] = np.array([thr, -1.0 if sp is None else sp])
    np.savez_compressed(os.path.join(OUT, "g14b_bsa_cdf.npz"), **out)
    print("g14b", {k: v.tolist() for k, v in out.items() if k.endswith("_lens")})


if __name__ == "__main__" and "bsa_cdf" in sys.argv[1:]:
    g_bsa_cdf()


def g_bsa_cdf_bf16(out_dir=OUT):
    """G14c: the reference's cdf / cdf+top-k COUNTS on BF16 block scores -- the dtype its bf16 model hands
    get_select_indices_cdf_from_score / _cdf_topk_from_score (bsa_interface.py:234-243, 253-266), where eager torch rounds score * scale,
    the softmax weights and every cumsum output to bf16.  Counts only: ties among bf16 weights make the sorted INDEX order arbitrary.
    Needs TORCHDYNAMO_DISABLE=1 (the functions are @torch.compile'd)."""
    _longcat_paths()
    from longcat_video.block_sparse_attention import bsa_interface as B

    g = torch.Generator().manual_seed(23)
    out = {}
    for name, (Hh, nq, nk, scale) in {"flat770": (2, 24, 770, 1.0), "mid770": (2, 24, 770, 6.0), "peaked770": (2, 24, 770, 20.0),
                                      "small96": (3, 16, 96, 4.0)}.items():
        score = (torch.randn(1, Hh, nq, nk, generator=g) * scale).bfloat16()
        out[f"{name}_score_bits"] = score[0].view(torch.int16).numpy().view(np.uint16)
        for thr, sp in ((0.3, None), (0.5, None), (0.9, None), (0.95, None), (0.3, 0.75), (0.9, 0.875)):
            if sp is None:
                _, lens = B.get_select_indices_cdf_from_score(score, thr, 1 / 128 ** 0.5)
            else:
                _, lens = B.get_select_indices_cdf_topk_from_score(score, sp, thr, 1 / 128 ** 0.5)
            out[f"{name}_lens_thr{thr}_sp{sp}"] = lens[0].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(out_dir, "g14c_bsa_cdf_bf16.npz"), **out)
    print("g14c", {k: (int(v.min()), int(v.max())) for k, v in out.items() if "_lens_" in k})


if __name__ == "__main__" and "bsa_cdf_bf16" in sys.argv[1:]:
    g_bsa_cdf_bf16()


# ------------------------------------------------------------------------------------------------------------
def g_bsa_triton(out_dir=OUT):
    """G18: the reference's block-sparse attention forward -- its Triton kernels (_attn_fwd_bsa_varlen_align, flash_attn_bsa_varlen_mask.py:
    174-285) launched by its own attn_fwd_bsa_varlen_triton / flash_attn_bsa / flash_attn_bsa_3d (bsa_interface.py:290-341, 535-560,
    612-659), unmodified, on CPU tensors through Triton's interpreter (TRITON_INTERPRET=1 must be set BEFORE triton is imported:
    @triton.jit then yields interpreted functions).  fp32 and fp16 tensors (the interpreter's bf16 path returns garbage).  Every 4th
    output row is stored; inputs are regenerated from tests/fakes.bsa_triton_inputs, their sums are stored to catch generator drift."""
    assert os.environ.get("TRITON_INTERPRET") == "1", "run with TRITON_INTERPRET=1 TORCHDYNAMO_DISABLE=1"
    _longcat_paths()
    from longcat_video.block_sparse_attention import bsa_interface as B
    from tests.fakes import BSA_TRITON_CASES, bsa_triton_inputs, bsa_varlen_lists

    out = {}
    for name, c in BSA_TRITON_CASES.items():
        q, k, v = (t.unsqueeze(0) for t in bsa_triton_inputs(name))
        if c["dtype"] == "f16":
            q, k, v = q.half(), k.half(), v.half()
        blk, scale = c["block"], 128 ** -0.5
        if name == "thw":
            o = B.flash_attn_bsa_3d(q, k, v, c["grid"], c["grid"], sparsity=c["sparsity"], chunk_3d_shape_q=list(c["chunk"]),
                                    chunk_3d_shape_k=list(c["chunk"]))
        elif name == "varlen":
            idx, lens = bsa_varlen_lists(name)
            o, lse = B.attn_fwd_bsa_varlen_triton(q, k, v, scale, idx.unsqueeze(0), lens.unsqueeze(0), blk, blk, 0.5)
            out[f"{name}_idx"], out[f"{name}_lens"] = idx.numpy(), lens.numpy()
        else:
            qc, kc = B.mean_pooling_compression(q, blk), B.mean_pooling_compression(k, blk)
            idx, lens = B.get_select_indices(qc, kc, c["sparsity"], None)
            o, lse = B.attn_fwd_bsa_varlen_triton(q, k, v, scale, idx, lens, blk, blk, c["sparsity"])
            assert torch.equal(o, B.flash_attn_bsa(q, k, v, blk, blk, c["sparsity"], None, scale))  # the autograd entry is the same path
            out[f"{name}_idx"], out[f"{name}_lens"] = idx[0].numpy(), lens[0].numpy()
        assert torch.isfinite(o.float()).all()
        out[f"{name}_out"] = o[0, :, ::4].float().numpy()
        out[f"{name}_insum"] = np.array([q.double().sum().item(), k.double().sum().item(), v.double().sum().item()])
    np.savez_compressed(os.path.join(out_dir, "g18_bsa_triton.npz"), **out)
    print("g18", {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__" and "bsa_triton" in sys.argv[1:]:
    g_bsa_triton()


# ------------------------------------------------------------------------------------------------------------
def warp_case(seed=4, H=48, W=64):
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (2.0 + 0.6 * np.sin(xx / 9.0) + 0.4 * np.cos(yy / 7.0) + 0.05 * g.standard_normal((H, W))).astype(np.float32)
    depth[(xx > 40) & (yy < 20)] = 1.1          # a near object: occlusions and disocclusions
    depth[5:8, 5:9] = np.nan                     # invalid depth
    image = g.random((H, W, 3)).astype(np.float32)
    K = np.array([[55.0, 0, W / 2 - 0.5], [0, 55.0, H / 2 - 0.5], [0, 0, 1]])
    th = 0.1
    E = np.eye(4)
    E[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    E[:3, 3] = [0.05, -0.02, 0.1]
    return image, depth, K, E


def g_warp():
    """G16: vggt/modules/utils_warp.py warp_single_img, unmodified, fill_cracks=False (the OpenCV-free part), two camera paths."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "refshim_cv2"))
    sys.path.insert(0, "/root/reference/vggt")
    from modules import utils_warp as UW

    image, depth, K, E = warp_case()
    out = dict(image=image, depth=depth, K=K, E=E)
    for name, (direction, deg, fn) in {"right": ("right", 12.0, UW.get_look_right_camera_seq), "forward": ("forward", 30.0, UW.get_look_forward_camera_seq)}.items():
        imgs, masks, infos = UW.warp_single_img(E[:3], K, torch.from_numpy(image).permute(2, 0, 1), torch.from_numpy(depth), depth_conf=None,
                                                direction=direction, degree=deg, frame_num=5, fill_cracks=False)
        mean_depth = np.nanmean(depth[~np.isnan(depth) & (depth > 0)])
        cams = fn(E.copy(), deg, 5, mean_depth)
        out[f"{name}_cams"] = np.stack(cams)
        out[f"{name}_imgs"], out[f"{name}_masks"] = np.stack(imgs), np.stack(masks)
        print("g16", name, out[f"{name}_imgs"].shape, float(np.stack(masks)[1:].mean()))
    np.savez_compressed(os.path.join(OUT, "g16_warp.npz"), **out)


if __name__ == "__main__" and "warp" in sys.argv[1:]:
    g_warp()


def g_warp_cams():
    """G16b: the reference's ten camera-path generators through warp_single_img's dispatch (utils_warp.py:818-839)."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "refshim_cv2"))
    sys.path.insert(0, "/root/reference/vggt")
    from modules import utils_warp as UW

    _, _, _, E = warp_case()
    fns = {"up": (UW.get_look_up_camera_seq, 1), "down": (UW.get_look_up_camera_seq, -1), "right": (UW.get_look_right_camera_seq, 1),
           "left": (UW.get_look_right_camera_seq, -1), "forward": (UW.get_look_forward_camera_seq, 1),
           "backward": (UW.get_look_backward_camera_seq, 1), "up_pan": (UW.get_up_pan_camera_seq, 1), "down_pan": (UW.get_down_pan_camera_seq, 1),
           "left_pan": (UW.get_left_pan_camera_seq, 1), "right_pan": (UW.get_right_pan_camera_seq, 1)}
    out = {"E": E}
    for name, (fn, sgn) in fns.items():
        out[name] = np.stack(fn(E.copy(), sgn * 17.0, 6, 2.3))
    np.savez_compressed(os.path.join(OUT, "g16b_warp_cams.npz"), **out)
    print("g16b", len(out) - 1, "paths")


if __name__ == "__main__" and "warp_cams" in sys.argv[1:]:
    g_warp_cams()


# ------------------------------------------------------------------------------------------------------------
def small_cracks_case(seed=9, H=40, W=56):
    """A nearly empty splatted view (<= 100 valid pixels, the trigger of utils_warp.py:957-962 else-branch) built to exercise BOTH steps of
    fill_small_cracks: a few compact islands of valid pixels with 1-4 pixel holes inside them, and depth steps across some holes."""
    g = np.random.default_rng(seed)
    img = g.random((H, W, 3)).astype(np.float32)
    mask = np.zeros((H, W), dtype=np.uint8)
    for (y, x, h, w) in ((3, 4, 5, 6), (20, 30, 5, 5), (30, 8, 4, 6), (10, 44, 4, 5)):       # 99 pixels before the holes are cut
        mask[y:y + h, x:x + w] = 1
    # a 1 x 3 and a 3 x 1 line (with min_valid_neighbors 7 step 1 fills the ends, step 2 then the middle, which by then has 8 valid
    # neighbours), a single pixel, a 2 x 2 block
    for (y, x) in ((5, 5), (5, 6), (5, 7), (21, 32), (22, 32), (23, 32), (31, 10), (11, 45), (11, 46), (12, 45), (12, 46)):
        mask[y, x] = 0
    depth = (2.0 + 0.02 * g.standard_normal((H, W))).astype(np.float32)
    depth[20:26, 33:] += 0.5        # a depth step beside the vertical line: three of its neighbours fail the depth test
    return img, mask, depth


def g_small_cracks():
    """G23: vggt/modules/utils_warp.py fill_small_cracks (:386-455), unmodified, on a <= 100-pixel view, with and without a confidence map.
    `import cv2` is served by a stand-in whose filter2D / morphologyEx are oracle/crackfill.py's restatements (OpenCV is absent: step 1's two
    stencils stay unpinned); everything the golden adds -- ndimage.label, the size rules, the sequential depth-guided fill of step 2 -- is the
    reference's own numpy / scipy code."""
    import types
    from oracle import crackfill as ocf
    cv2 = types.ModuleType("cv2")
    cv2.MORPH_CLOSE = 3
    cv2.morphologyEx = lambda m, op, k: ocf.close3(m)
    cv2.filter2D = lambda im, dd, k, **kw: ocf.filter2d(im, k, "reflect" if kw.get("borderType") else "reflect101")
    sys.modules["cv2"] = cv2
    sys.path.insert(0, "/root/reference/vggt")
    from modules import utils_warp as UW
    img, mask, depth = small_cracks_case()
    out = dict(img=img, mask=mask, depth=depth)
    # with the shipped parameters (min_valid_neighbors 3 or 2) step 1 already fills every hole step 2 could reach (a 4-connected hole of <= 4
    # pixels is enclosed by valid pixels, so each of its pixels has >= 3 valid neighbours); stricter counts make step 2 do the work
    from tests.cases import SMALL_CRACK_CASES
    for name, (has_conf, mcs, mvn, thr) in SMALL_CRACK_CASES.items():
        conf = np.ones_like(depth) if has_conf else None
        fi, fm = UW.fill_small_cracks(img, mask, depth, depth_conf=conf, depth_threshold=thr, max_crack_size=mcs, min_valid_neighbors=mvn)
        out[f"{name}_img"], out[f"{name}_mask"] = fi, fm
        f1, m1 = ocf.fill_small_cracks(img, mask, mvn)
        print("g23", name, int(fm.sum() - mask.sum()), "pixels filled,", int(fm.sum() - m1.sum()), "of them by step 2")
    np.savez_compressed(os.path.join(OUT, "g23_fill_small_cracks.npz"), **out)


if __name__ == "__main__" and "small_cracks" in sys.argv[1:]:
    g_small_cracks()
