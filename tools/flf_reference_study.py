"""FLF gate reproducibility ON THE IMPORTED REFERENCE (VERDICT r2 "Next" #2).  Build-container only (needs /root/reference).

DESIGN section 4b claims that at schedule length the FLF gate (SCHED:338-437: a discrete choice over 16 nearly tied similarities) is
not reproducible between two arithmetically close runs, and showed it on the builder's oracle (8 vs 128 threads).  This tool runs the
SAME job (tools/vae_precision_study.py: 20-step schedule, 15 guided steps x 2 rounds = 31 decode -> blend -> encode round trips,
17 x 128 x 128, CFG 4, omega 4) through the UNMODIFIED reference:

    pipeline   utils/pipeline_wan_i2v_clean.py  WanImageToVideoPipeline.__call__          (imported through tools/refshim)
    scheduler  utils/scheduling_unipc_multistep_clean.py  UniPCMultistepScheduler + VideoMotionPCASelector
    VAE        diffusers' AutoencoderKLWan as vendored at longcat_video/modules/autoencoder_kl_wan.py (real config, fp32)
    DiT        the in-tree twin wan/modules/model.py WanModel (d = 1024, 4 layers, fp32; flash_attention -> the file's own SDPA fallback)

with `torch.set_num_threads(n)` for n in --threads, and records every gate: step, the 16 similarities, the selected channels, the
decision margin; plus per-step latents and sampled frames (fp16).  --flow tdiff = the branch the reference executes here (cv2 absent ->
ImportError at SCHED:159-161 caught at :390-392): 100 % reference code.  --flow farneback = the deployed branch with `import cv2`
served by a stand-in module whose calcOpticalFlowFarneback is oracle/farneback.py (builder's restatement of OpenCV: the thread-count
question is about what happens UPSTREAM of the flow, so the stand-in does not decide the answer, but the run is labelled).

    python tools/flf_reference_study.py --threads 8 --flow tdiff --out tests/golden/g19_flf_reference_tdiff_t8.npz
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/wan_for_worldforge"
sys.path.insert(0, os.path.join(ROOT, "tools", "refshim"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from make_goldens import _ImgEnc, _ImgProc, _load_akw, _load_wan_module  # noqa: E402
from vae_precision_study import FIXTURE_FRAMES, decision_margin, make_inputs  # noqa: E402


def install_cv2_standin():
    """`import cv2` inside SCHED:158 -> a module with the three names that code path touches, served by oracle/farneback.py."""
    from oracle import farneback as ofb
    m = types.ModuleType("cv2")
    m.COLOR_RGB2GRAY = 7

    def cvtColor(img, code):
        assert code == m.COLOR_RGB2GRAY
        # the reference replicates ONE channel three times (SCHED:386-388): OpenCV's fixed-point luma of (v, v, v) is v exactly
        assert (img[..., 0] == img[..., 1]).all() and (img[..., 0] == img[..., 2]).all()
        return np.ascontiguousarray(img[..., 0])

    def calcOpticalFlowFarneback(prev, nxt, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        assert (pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags) == (0.5, 3, 15, 3, 5, 1.2, 0)
        return ofb.calc_optical_flow_farneback(prev, nxt)

    m.cvtColor, m.calcOpticalFlowFarneback = cvtColor, calcOpticalFlowFarneback
    sys.modules["cv2"] = m


class TwinDiT:
    """diffusers call protocol (PIPE:593-600) in front of the in-tree twin WanModel (model.py:493-582)."""

    def __init__(self, wmodel, W, dim, ffn, heads, layers):
        self.dtype = torch.bfloat16
        self.config = types.SimpleNamespace(patch_size=(1, 2, 2))
        self.m = wmodel.WanModel(model_type="i2v", in_dim=36, dim=dim, ffn_dim=ffn, freq_dim=256, text_dim=64, out_dim=16, num_heads=heads,
                                 num_layers=layers)
        self.m.load_state_dict(W, strict=True)
        self.m.eval()
        self.calls = 0

    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None, attention_kwargs=None,
                 return_dict=False):
        self.calls += 1
        x = hidden_states[0].float()
        T, h, w = x.shape[1:]
        with torch.no_grad():
            o = self.m([x[:16]], timestep.reshape(-1)[:1], [encoder_hidden_states[0].float()], seq_len=T * (h // 2) * (w // 2),
                       clip_fea=encoder_hidden_states_image.float(), y=[x[16:]])[0]
        return (o.unsqueeze(0).to(self.dtype),)


def build(dim, layers):
    from oracle import dit as odit
    from oracle import vae as ovae
    from worldforge_amd.vae import diffusers_key_map

    heads, ffn = dim // 128, 2 * dim
    ocfg = odit.DiTConfig(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64)
    Wd = odit.random_weights(ocfg, seed=3)
    Wd = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in Wd.items()}
    wattn, wmodel = _load_wan_module("attention"), _load_wan_module("model")
    orig_attention = wattn.attention
    wmodel.flash_attention = lambda q, k, v, k_lens=None, window_size=(-1, -1): orig_attention(
        q, k, v, k_lens=k_lens, window_size=window_size, fa_version=None, dtype=torch.float32)
    dit = TwinDiT(wmodel, Wd, dim, ffn, heads, layers)
    akw = _load_akw()
    vae = akw.AutoencoderKLWan()
    vae.eval()
    sd = vae.state_dict()
    Wv = ovae.random_weights(seed=4)
    inv = {v: k for k, v in diffusers_key_map().items()}
    new = {}
    for k, v in Wv.items():
        base, _, leaf = k.rpartition(".")
        new[f"{inv[base]}.{leaf}"] = v.reshape(sd[f"{inv[base]}.{leaf}"].shape)
    vae.load_state_dict(new, strict=True)
    return dit, vae


def run(threads, flow, dim=1024, layers=4, Fr=17, H=128, Wd=128, steps=20, guide=15):
    torch.set_num_threads(threads)
    if flow == "farneback":
        install_cv2_standin()
    from utils.pipeline_wan_i2v_clean import WanImageToVideoPipeline
    from utils import scheduling_unipc_multistep_clean as S

    dit, vae = build(dim, layers)
    image, ref, mask, text, neg, img = make_inputs(Fr, H, Wd)
    sch = S.UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=3.0)
    pipe = WanImageToVideoPipeline(tokenizer=None, text_encoder=None, image_encoder=_ImgEnc(img), image_processor=_ImgProc(),
                                   transformer=dit, vae=vae, scheduler=sch)
    gates, sims_box = [], []
    Sel = S.VideoMotionPCASelector
    orig_sel, orig_corr = Sel.select_motion_related_channels, Sel._compute_channel_correlations

    def corr(self, *a, **k):
        s = orig_corr(self, *a, **k)
        sims_box.append([float(v) for v in s])
        return s

    def sel(self, *a, **k):
        n = len(sims_box)
        ch = orig_sel(self, *a, **k)
        step = k.get("current_step", None)
        gates.append((int(step) if step is not None else -1, [int(c) for c in ch], sims_box[-1] if len(sims_box) > n else None))
        return ch

    Sel.select_motion_related_channels, Sel._compute_channel_correlations = sel, corr
    lat = []

    def cb(p, i, t, kw):
        lat.append(kw["latents"].detach().float().clone())
        return {}

    t0 = time.time()
    try:
        out = pipe(image=image, prompt=None, negative_prompt=None, height=H, width=Wd, num_frames=Fr, num_inference_steps=steps,
                   guidance_scale=4.0, generator=torch.manual_seed(42), prompt_embeds=text, negative_prompt_embeds=neg,
                   output_type="np", video_ref=ref, mask=mask, guided=True, resample_steps=2, guide_steps=guide, omega=4.0,
                   omega_resample=4.0, resample_round=guide, use_pca_channel_selection=True, static=True, callback_on_step_end=cb)
    finally:
        Sel.select_motion_related_channels, Sel._compute_channel_correlations = orig_sel, orig_corr
    frames = np.asarray(out.frames, dtype=np.float32)[0]  # [F, H, W, 3]
    job = dict(dit=f"d{dim} x {layers} layers x {dim // 128} heads", frames=Fr, height=H, width=Wd, steps=steps, guided_steps=guide,
               round_trips=2 * guide + 1, flow_backend=flow)
    meta = dict(job=job, threads=threads, seconds=round(time.time() - t0, 1), torch=torch.__version__, dit_calls=dit.calls,
                source="unmodified reference pipeline + scheduler + vendored AutoencoderKLWan + in-tree WanModel twin"
                       + ("; cv2 served by oracle/farneback.py" if flow == "farneback" else "; cv2 absent -> the reference's own temporal-difference branch"))
    return frames, lat, gates, meta


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--flow", default="tdiff", choices=["tdiff", "farneback"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--guide", type=int, default=15)
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    frames, lat, gates, meta = run(a.threads, a.flow, a.dim, a.layers, a.frames, a.size, a.size, a.steps, a.guide)
    fidx = [i for i in FIXTURE_FRAMES if i < frames.shape[0]]
    margins = [(st, decision_margin(sm, st)) for st, _, sm in gates]
    np.savez_compressed(a.out, meta=json.dumps(meta), job=json.dumps(meta["job"]),
                        frames=frames[fidx].astype(np.float16),  # [n, H, W, 3] in [0, 1], the layout of the oracle fixture g17
                        frame_idx=np.array(fidx), latents=np.stack([x.numpy() for x in lat]).astype(np.float16),
                        flf_lists=json.dumps([(st, ch) for st, ch, _ in gates]), flf_sims=json.dumps([sm for _, _, sm in gates]),
                        flf_margins=json.dumps(margins))
    print(json.dumps(dict(meta=meta, gates=[(st, ch) for st, ch, _ in gates], margins=margins)))
