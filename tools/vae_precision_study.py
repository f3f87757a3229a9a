"""Schedule-length VAE precision study (VERDICT r1 "Next" #2).

The reference runs the VAE in fp32 (infer_worldforge.py:185-189) and a guided job pushes the latents through 31 decode -> blend
-> encode round trips (scheduling_unipc_multistep_clean.py:1281-1385, twice per guided step for guide = round = 15, plus the final
decode).  The product's default VAE rounds every matrix-core operand to bf16.  This tool runs ONE guided job (IRR + FLF + DSG,
CFG 4, 20-step schedule with 15 guided steps x 2 rounds, so FLF swaps channels from step 6 on and takes the > 10 branch) three
ways with identical weights, seeds and inputs:

    A  HIP path, VAE precision="bf16"      (the bench configuration)
    B  HIP path, VAE precision="fp32"      (three-term split operands, fp32-class contractions)
    C  CPU oracle, fp32                    (the parity target; optional: --oracle)

and reports PSNR of the final frames pairwise, the per-step relative distance of the latents (the dB-vs-round-trips curve) and the
FLF channel lists of every gate.  A vs B isolates the VAE's operand precision (everything else is bit-identical HIP arithmetic);
B vs C is what the rest of the path (bf16 DiT activations) contributes.

Because the FLF gate is a DISCRETE decision on 16 nearly tied similarities, two arithmetically close runs can swap different channels
and then diverge (a swapped channel is a large change of x0).  The tool therefore reports both the free-running comparison (with the
decision margin of every gate) and the comparison with the parity target's decisions replayed (scheduler.flf_replay).

Usage:  python tools/vae_precision_study.py [--oracle [--save-fixture F.npz] | --fixture F.npz] [--frames 17 --height 128 --width 128
                                             --steps 20 --guide 15] [--out FILE]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


FIXTURE_FRAMES = [0, 4, 8, 12, 16]  # the recorded oracle result keeps these frames (fp16) + every step's latents (fp16)


def psnr(a, b):
    mse = ((a.float() - b.float()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-12))


def db(a, b):
    """20 log10(|b| / |a - b|): signal-to-difference ratio of two latent tensors."""
    return 20 * math.log10(b.float().norm().item() / max((a.float() - b.float()).norm().item(), 1e-12))


def make_inputs(Fr, H, Wd):
    g = torch.Generator().manual_seed(7)
    image = torch.rand(3, H, Wd, generator=g)
    # a smooth moving pattern (FLF compares motion): low-frequency field translating over the frames
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(Wd).float(), indexing="ij")
    ph = torch.rand(3, 4, generator=g) * 6.28
    frames = []
    for f in range(Fr):
        ch = [0.5 + 0.25 * torch.sin((xx + 3 * f) / 9.0 + ph[c, 0]) * torch.cos((yy - 2 * f) / 7.0 + ph[c, 1])
              + 0.2 * torch.sin((xx - yy + 5 * f) / 13.0 + ph[c, 2]) for c in range(3)]
        frames.append(torch.stack(ch))
    ref = torch.stack(frames, dim=1).unsqueeze(0).clamp(0, 1) + 0.02 * torch.rand(1, 3, Fr, H, Wd, generator=g)
    ref = ref.clamp(0, 1)
    ref[:, :, 0] = image
    xs = torch.arange(Wd).view(1, 1, 1, 1, Wd).float()
    fr = torch.arange(Fr).view(1, 1, Fr, 1, 1).float() / max(Fr - 1, 1)
    mask = (xs < Wd * (1 - 0.35 * fr)).float().expand(1, 1, Fr, H, Wd).contiguous()
    text = (torch.randn(1, 24, 64, generator=g) * 0.5).to(torch.bfloat16)
    neg = (torch.randn(1, 24, 64, generator=g) * 0.5).to(torch.bfloat16)
    img = torch.randn(1, 257, 1280, generator=g).to(torch.bfloat16)
    return image, ref, mask, text, neg, img


def run_hip(precision, Wd_, Wv, ocfg, inputs, Fr, H, Wd, steps, guide, flow_backend, replay=None):
    from worldforge_amd import dit
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    dev = torch.device("cuda:0")
    image, ref, mask, text, neg, img = inputs
    cfg = dit.DiTConfig(dim=ocfg.dim, ffn_dim=ocfg.ffn_dim, num_heads=ocfg.num_heads, num_layers=ocfg.num_layers, text_dim=64)
    model = dit.WanTransformer3DModel(cfg, dev).load_state_dict(Wd_)
    vae = AutoencoderKLWan(dev, precision=precision).load_state_dict(Wv)
    sch = UniPCMultistepScheduler(flow_shift=3.0, flow_backend=flow_backend)
    sch.flf_log = []
    sch.flf_replay = dict(replay) if replay is not None else None
    lat = []
    pipe = WanImageToVideoPipeline(model, vae, sch, device=dev)

    def on_step_end(_pipe, i, t, kw):  # the reference's own per-step callback (PIPE:716-723)
        lat.append(kw["latents"].detach().float().cpu().clone())
        return {}

    out = pipe(image=image, height=H, width=Wd, num_frames=Fr, num_inference_steps=steps, guidance_scale=4.0,
               generator=torch.manual_seed(42), prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img, output_type="np",
               video_ref=ref, mask=mask, static=True, guided=True, resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0,
               resample_round=guide, use_pca_channel_selection=True, callback_on_step_end=on_step_end)
    return torch.from_numpy(out.frames)[0], lat, [(e[0], e[1]) for e in sch.flf_log], [e[2] for e in sch.flf_log]


def run_oracle(Wd_, Wv, ocfg, inputs, Fr, H, Wd, steps, guide, flow_backend):
    from oracle import dit as odit
    from oracle import inject as oinject
    from oracle import sampler as osampler
    from oracle import vae as ovae

    image, ref, mask, text, neg, img = inputs
    gen = torch.manual_seed(42)
    T = (Fr - 1) // 4 + 1
    lat = torch.randn((1, 16, T, H // 8, Wd // 8), generator=gen, dtype=torch.float32)
    dec = lambda z: ovae.decode(Wv, z)  # noqa: E731
    enc = lambda x: ovae.encode_mode(Wv, x)  # noqa: E731
    cond = osampler.prepare_condition((2.0 * image - 1.0).unsqueeze(0), Fr, enc, ovae.MEAN, ovae.STD)
    scfg = osampler.SamplerConfig(num_inference_steps=steps, guidance_scale=4.0, flow_shift=3.0, flow_backend=flow_backend, guided=True,
                                  resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0, resample_round=guide,
                                  use_pca_channel_selection=True)
    log = []
    orig = oinject.select_motion_related_channels

    sims_log = []
    orig_sims = oinject.channel_similarities

    def logged_sims(pred, enc_, flow_backend="tdiff"):
        sm = orig_sims(pred, enc_, flow_backend)
        sims_log.append([float(v) for v in sm])
        return sm

    def logged(pred, enc_, current_step, flow_backend="tdiff"):
        n = len(sims_log)
        ch = orig(pred, enc_, current_step, flow_backend=flow_backend)
        log.append((int(current_step), list(ch), sims_log[-1] if len(sims_log) > n else None))
        return ch

    oinject.select_motion_related_channels = logged
    oinject.channel_similarities = logged_sims

    def tr(x, t, ctx, im):
        v = odit.forward(Wd_, ocfg, x[0].float(), t.reshape(-1)[0], ctx[0].float(), im[0].float())
        return v.unsqueeze(0).to(torch.bfloat16)

    trace = []
    try:
        with torch.no_grad():
            out = osampler.run(scfg, latents=lat, condition=cond, transformer=tr, prompt_embeds=text, negative_prompt_embeds=neg,
                               image_embeds=img, video_ref=ref, mask=mask, decode=dec, encode_mode=enc, mean=ovae.MEAN, std=ovae.STD,
                               generator=gen, trace=trace)
            frames = osampler.decode_final(out, dec, ovae.MEAN, ovae.STD)[0]
    finally:
        oinject.select_motion_related_channels = orig
        oinject.channel_similarities = orig_sims
    lats = [e[2].float() for e in trace if e[0] == "latents"]
    return frames, lats, [(s, c) for s, c, _ in log], [sm for _, _, sm in log]


def decision_margin(sims, step):
    """How far the FLF decision at this gate is from flipping, in similarity units.  Every rule of SCHED:408-437 selects a PREFIX of the
    ascending-sorted similarities (the lowest 1; or all below mean - 0.625 std, clamped to 2..6): the decision changes when the order
    across the prefix boundary changes (gap between the last selected and the first unselected) or, when the prefix length comes from
    the threshold, when the threshold crosses a similarity."""
    import numpy as np
    if sims is None or step <= 5:
        return None
    c = np.sort(np.asarray(sims, dtype=np.float64))
    if step <= 10:
        return float(c[1] - c[0])
    thr = c.mean() - 0.625 * c.std()
    n = int((c < thr).sum())
    k = min(max(n, 2), 6)
    gap = float(c[k] - c[k - 1])
    if 2 <= n <= 6:
        gap = min(gap, float(min(abs(thr - c[n - 1]), abs(c[n] - thr))))
    return gap


def study(dim=1024, ffn_dim=2048, heads=8, layers=4, Fr=17, H=128, Wd=128, steps=20, guide=15, flow_backend="farneback",
          with_oracle=False, verbose=True, fixture=None, save_fixture=None, oracle_only=False, skip_bf16=False):
    """fixture: an .npz written by an earlier `--oracle --save-fixture` run (oracle frames / latents / gate decisions for exactly
    this job) used instead of running the CPU oracle again."""
    import numpy as np
    from oracle import dit as odit
    from oracle import vae as ovae

    ocfg = odit.DiTConfig(dim=dim, ffn_dim=ffn_dim, num_heads=heads, num_layers=layers, text_dim=64)
    Wd_ = odit.random_weights(ocfg, seed=3)
    Wd_ = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in Wd_.items()}
    Wv = ovae.random_weights(seed=4)
    inputs = make_inputs(Fr, H, Wd)
    job = dict(dit=f"d{dim} x {layers} layers x {heads} heads", frames=Fr, height=H, width=Wd, steps=steps, guided_steps=guide,
               round_trips=2 * guide + 1, flow_backend=flow_backend)
    res = {"config": job}
    args = (Wd_, Wv, ocfg, inputs, Fr, H, Wd, steps, guide, flow_backend)
    if oracle_only:  # CPU only: record the oracle's result for this job (no GPU needed)
        t0 = time.time()
        orc = run_oracle(*args)
        np.savez_compressed(save_fixture, job=json.dumps(job), frames=orc[0].numpy().astype(np.float16)[FIXTURE_FRAMES], frame_idx=np.array(FIXTURE_FRAMES),
                            latents=np.stack([x.numpy() for x in orc[1]]).astype(np.float16), flf_lists=json.dumps(orc[2]), flf_sims=json.dumps(orc[3]))
        print(f"oracle: {time.time() - t0:.0f} s -> {save_fixture}; gates {orc[2]}")
        return res
    t0 = time.time()
    fb, lb, cb, sb = run_hip("bf16x3", *args)
    fa, la, ca, sa = (fb, lb, cb, sb) if skip_bf16 else run_hip("bf16", *args)   # skip_bf16: the "bf16vae" keys repeat the default mode
    res["hip_s"] = time.time() - t0
    res["free_running"] = {
        "psnr_bf16vae_vs_fp32vae_db": psnr(fa, fb), "flf_gates": len(ca), "flf_swapping_gates": sum(1 for _, c in cb if c),
        "flf_lists_bf16vae": ca, "flf_lists_fp32vae": cb, "flf_same_bf16_vs_fp32vae": ca == cb,
        "gate_margin_fp32vae": [(st, decision_margin(sm, st)) for (st, _), sm in zip(cb, sb)],
        "gate_max_sim_delta_bf16_vs_fp32vae": [(st, None if x is None or y is None else float(np.abs(np.asarray(x) - np.asarray(y)).max()))
                                               for (st, _), x, y in zip(cb, sa, sb)],
        "latent_db_bf16_vs_fp32vae_per_step": [round(db(a, b), 1) for a, b in zip(la, lb)]}
    orc, fidx = None, None
    sel = lambda f: f if fidx is None else f[fidx]  # noqa: E731  (a recorded oracle result keeps a subset of the frames)
    if fixture is not None:
        z = np.load(fixture, allow_pickle=False)
        assert json.loads(str(z["job"])) == json.loads(json.dumps(job)), "fixture was recorded for a different job"
        lists = json.loads(str(z["flf_lists"]))
        fidx = [int(i) for i in z["frame_idx"]]
        orc = (torch.from_numpy(z["frames"].astype(np.float32)), [torch.from_numpy(x.astype(np.float32)) for x in z["latents"]],
               [(int(st), list(c)) for st, c in lists], json.loads(str(z["flf_sims"])))
    elif with_oracle:
        t0 = time.time()
        orc = run_oracle(*args)
        res["oracle_s"] = time.time() - t0
        if save_fixture:
            np.savez_compressed(save_fixture, job=json.dumps(job), frames=orc[0].numpy().astype(np.float16)[FIXTURE_FRAMES], frame_idx=np.array(FIXTURE_FRAMES),
                                latents=np.stack([x.numpy() for x in orc[1]]).astype(np.float16), flf_lists=json.dumps(orc[2]), flf_sims=json.dumps(orc[3]))
    if orc is not None:
        fc, lc, cc, sc = orc
        res["free_running"].update({
            "psnr_bf16vae_vs_oracle_db": psnr(sel(fa), fc), "psnr_fp32vae_vs_oracle_db": psnr(sel(fb), fc), "flf_lists_oracle": cc,
            "flf_same_bf16vae_vs_oracle": ca == cc, "flf_same_fp32vae_vs_oracle": cb == cc,
            "gate_margin_oracle": [(st, decision_margin(sm, st)) for (st, _), sm in zip(cc, sc)],
            "gate_max_sim_delta_fp32vae_vs_oracle": [(st, None if x is None or y is None else float(np.abs(np.asarray(x) - np.asarray(y)).max()))
                                                     for (st, _), x, y in zip(cc, sb, sc)],
            "latent_db_bf16vae_vs_oracle_per_step": [round(db(a, c), 1) for a, c in zip(la, lc)],
            "latent_db_fp32vae_vs_oracle_per_step": [round(db(b, c), 1) for b, c in zip(lb, lc)]})
    # the same job with the gate decisions of the parity target replayed: what the ARITHMETIC of the path contributes
    target = dict(orc[2]) if orc is not None else dict(cb)
    rb, lrb, _, _ = run_hip("bf16x3", *args, replay=target)
    ra, lra = (rb, lrb) if skip_bf16 else run_hip("bf16", *args, replay=target)[:2]
    rep = {"decisions_from": ("fixture" if fixture is not None else "oracle") if orc is not None else "fp32-class VAE run", "psnr_bf16vae_vs_fp32vae_db": psnr(ra, rb),
           "latent_db_bf16_vs_fp32vae_per_step": [round(db(a, b), 1) for a, b in zip(lra, lrb)]}
    if orc is not None:
        rep.update({"psnr_bf16vae_vs_oracle_db": psnr(sel(ra), orc[0]), "psnr_fp32vae_vs_oracle_db": psnr(sel(rb), orc[0]),
                    "latent_db_bf16vae_vs_oracle_per_step": [round(db(a, c), 1) for a, c in zip(lra, orc[1])],
                    "latent_db_fp32vae_vs_oracle_per_step": [round(db(b, c), 1) for b, c in zip(lrb, orc[1])]})
    res["decisions_replayed"] = rep
    if verbose:
        print(json.dumps(res, indent=None, default=str))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--guide", type=int, default=15)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--flow-backend", default="farneback")
    ap.add_argument("--out", default=None)
    ap.add_argument("--save-fixture", default=None, help="with --oracle: write the oracle's frames / latents / gate decisions (.npz)")
    ap.add_argument("--oracle-only", action="store_true", help="CPU only: run the oracle for this job and write --save-fixture")
    ap.add_argument("--fixture", default=None, help="use a recorded oracle result instead of running the CPU oracle")
    a = ap.parse_args()
    r = study(dim=a.dim, ffn_dim=2 * a.dim, heads=a.dim // 128, layers=a.layers, Fr=a.frames, H=a.height, Wd=a.width, steps=a.steps,
              guide=a.guide, flow_backend=a.flow_backend, with_oracle=a.oracle, fixture=a.fixture, save_fixture=a.save_fixture,
              oracle_only=a.oracle_only)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(r, f, indent=1, default=str)
