"""bench.py -- denoising steps/s of the IRR/FLF/DSG guided sampler (Wan2.1-I2V-14B, 81 frames x 480x832) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   ->  ONE JSON line on rank 0.
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the DiT token sequence is sharded over ranks
(K/V all-gather over RCCL, see worldforge_amd/parallel.py) -- "strong" scaling: the job (one video) is fixed.

Workload (BASELINE.json configs[1]): synthetic 81 x 480 x 832 video -> latents [1,16,21,60,104], L = 32 760 tokens,
random-init Wan2.1-I2V-14B DiT (40 layers, d = 5120, 40 heads, FFN 13824; bf16) and random-init real-config VAE (fp32 I/O),
CFG 4 (two DiT forwards per evaluation), 50-step schedule, guide_steps = resample_round = 15, resample_steps = 2, omega = 4,
FLF on.  A "step" is one outer iteration of PIPE:563.  The timed window holds K consecutive steps of that schedule entered at
step 15 - W - n_g so that n_g = max(1, round(0.3 K)) of them are guided steps (4 DiT forwards + 2 VAE decodes + 2 VAE encodes
+ FLF + DSG) and the rest are plain steps (2 DiT forwards) -- the 15:35 mix of the 50-step job.  Nothing is skipped or cached
inside the timed region; inputs are resident in HBM before it starts.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS_BF16 = 2500.0  # dense, MI355X_MICROARCH.md
# PMC side fields of the roofline object (HBM traffic per launch, MFMA pipe utilisation, effective clock) come from SEPARATE rocprofv3
# --pmc passes over tools/attn_once.py (the guide forbids mixing counters with the timed run); their summary is the tracked file below,
# written by tools/pmc_summary.py --json.  They are emitted only when this run launches the very kernel / shape the file describes.
ATTN_PMC_FILE = os.path.join(ROOT, "profiles", "attn_pmc_latest.json")



BOX_CALIB_REFERENCE_TFLOPS = 1800.0  # the bare-MFMA rate `value_normalised` is quoted at (round 3's lab box: 1818 on N(0,1) operands)
# How the headline workload (C2 on one GPU) follows two in-run speed proxies, FITTED on round 5's ten 1-GPU lines from eight boxes (value
# 0.2440 ... 0.2549 steps/s: a 4.3 % spread): the bare-MFMA calibration stream around the window (1751 ... 1844 TFLOP/s) and
# the average launch time of the self-attention kernel inside the window (14.95 ... 15.65 ms).  Least squares in log space:
#     value ~ calib^0.34 x attn_ms^-0.57      -> value_normalised = value x (1800 / calib)^0.34 x (attn_ms / 15.0)^0.57, spread 0.59 %.
# Either proxy alone leaves 2.0 % (calibration, best exponent 0.75; a proportional correction 2.2 %) or 1.2 % (attention time, exponent 0.8):
# the bare stream sees the socket's power-limited matrix clock, the attention time the box's behaviour under the real instruction mix.
BOX_CALIB_EXPONENT = 0.34
BOX_ATTN_REFERENCE_MS = 15.0
BOX_ATTN_EXPONENT = 0.57
_CALIB = {}


def box_calib_tflops(device, launches: int = 7, iters: int = 400_000):
    """What THIS box sustains on the matrix pipe alone, right now: a fixed register-only stream of v_mfma_f32_32x32x16_bf16 on N(0,1)
    operands (wf_calib_mfma: the round-3 energy lab's variant 0), `launches` launches of ~0.12 s back to back, HIP events, median of
    the last four (the first ones ride the clock down to the power-limited steady state).  The boxes of the pool differ by up to 6 % on
    the same binary; this is the in-run proxy that makes lines from different boxes comparable -- a diagnostic."""
    import ctypes
    from worldforge_amd import _ffi, ops
    if "src" not in _CALIB:
        g = torch.Generator(device=device).manual_seed(7)
        _CALIB["src"] = torch.randn(1 << 19, generator=g, device=device).to(torch.bfloat16)   # 1 MiB
        _CALIB["sink"] = torch.zeros(16, device=device)
    flop = ctypes.c_double(0.0)
    evs = []
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.call("wf_calib_mfma", _CALIB["src"].data_ptr(), _CALIB["sink"].data_ptr(), iters, ctypes.byref(flop), ops.stream())
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[-4:])
    return flop.value / (0.5 * (ms[1] + ms[2]) * 1e-3) / 1e12


def _transport() -> str:
    """What the collectives of this run travel over: RCCL, or -- debug runs with every rank on one GPU -- gloo through the host."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return "gloo through the host: DEBUG transport, timings meaningless"
    return "RCCL"


def attn_pmc_fields(kernel: str, L: int, heads: int, world: int, simulated: bool):
    """-> dict of roofline side fields, or all-None when the tracked PMC summary does not describe this launch."""
    none = {"traffic": None, "traffic_source": None, "mfma_util_pmc": None, "clock_ghz_pmc": None}
    try:
        with open(ATTN_PMC_FILE) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return none
    if simulated or world != 1 or pmc.get("kernel") != kernel or pmc.get("tokens") != L or pmc.get("heads") != heads:
        return none
    return {"traffic": pmc.get("traffic_bytes_per_launch"),
            "traffic_source": f"{os.path.relpath(ATTN_PMC_FILE, ROOT)} <- {pmc.get('source')}: rocprofv3 --pmc passes of {pmc.get('kernel')} at this shape, "
                              "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE; separate passes, not collected during this run",
            "mfma_util_pmc": pmc.get("mfma_util"), "clock_ghz_pmc": pmc.get("clock_ghz")}


def tracked_body_ms(model, L: int, heads: int, n: int = 6):
    """Average launch time of the SAME self-attention on the DiT's own resident Q / K / V^T of the last layer, with the norm bounds withheld:
    the kernel then runs its max-tracking body -- what a checkpoint with larger q / k norms would select.  Measured after the timed region."""
    from worldforge_amd import dit as wdit
    ws = {k[0]: v for k, v in model._ws.items()}
    if not all(k in ws for k in ("qh", "kh", "vt", "ao")):
        return None
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wdit.attention(ws["qh"], ws["kh"], ws["vt"], ws["ao"], L, 0.0, nsplit=1)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[1:])
    return ms[len(ms) // 2]


def flf_gate_ms(sch, pipe_latent_shape, device):
    """GPU time of one FLF gate (SCHED:338-437) with either motion backend on latents of the job's shape: the like-for-like cost of the
    default Farneback branch against the golden-pinned temporal-difference branch (2 gates per guided step)."""
    from worldforge_amd import flf
    g = torch.Generator(device=device).manual_seed(5)
    a = torch.randn(pipe_latent_shape, generator=g, device=device)
    b = a + 0.3 * torch.randn(pipe_latent_shape, generator=g, device=device)
    out = {}
    for backend in ("farneback", "tdiff"):
        sel = flf.VideoMotionPCASelector(flow_backend=backend)
        ms = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sel.select_motion_related_channels(pred_original_sample=a, video_latents=b, mask=None, keep_channels=12, current_step=12,
                                               total_steps=50, use_optical_flow=True, static=True)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        out[backend] = sorted(ms[1:])[1]
    return out


def synthetic_inputs(F, H, W, device, seed=42):
    """SURVEY 8d synthetic inputs."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(3, H, W, generator=g)
    ref = torch.rand(1, 3, F, H, W, generator=g)
    ref[:, :, 0] = image
    xs = torch.arange(W).view(1, 1, 1, 1, W).float()
    fr = torch.arange(F).view(1, 1, F, 1, 1).float() / max(F - 1, 1)
    edge = W * (1 - 0.35 * fr)
    d = (edge - xs).clamp(min=0)
    mask = (torch.sin(math.pi / 2 * (d / 15).clamp(0, 1)) * (xs < edge)).expand(1, 1, F, H, W).contiguous()
    text = torch.randn(1, 512, 4096, generator=g) * 0.1
    text[:, 200:] = 0
    neg = torch.randn(1, 512, 4096, generator=g) * 0.1
    neg[:, 120:] = 0
    img_emb = torch.randn(1, 257, 1280, generator=g)
    bf = torch.bfloat16
    return image, ref.to(device), mask.to(device), text.to(bf).to(device), neg.to(bf).to(device), img_emb.to(bf).to(device)


def _median3(fn):
    """median of 3 runs after 1 warm-up."""
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1]


def cpu_baseline(L, frames, height, width):
    """BASELINE.md section 3 on a bounded sample (~15-25 s on the GPU box's host): the oracle (CPU port of the reference arithmetic,
    fp32, all host threads), median of 3 after a warm-up of
      (i)   one full-width DiT block (d = 5120, 40 heads, FFN 13824, text+image cross-attention) at the C1 token count L1 = 4524, and its
            self-attention core alone at L1 -- everything in a block except that core is linear in L;
      (ii)  the self-attention core with the config's TRUE key length: L1 query rows x L keys on 4 of the 40 heads (the core is linear
            in query rows and in heads, so this prices the L^2 term at the real L without the 40 x L x L score tensor);
      (iii) VAE encode + decode of a 5 x 96 x 96 clip (linear in pixel-frames);
    returns per-unit CPU seconds at the config's true sizes: one DiT forward (40 blocks) and one VAE decode + encode."""
    from oracle import dit as odit
    from oracle import vae as ovae

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    cfg = odit.DiTConfig(num_layers=1)
    L1 = 4524  # BASELINE config 1: 9 frames of 464 x 832
    W = odit.random_weights(cfg, seed=1)
    f, h, w = 3, 29, 52
    tok = torch.randn(L1, cfg.dim)
    e0 = torch.randn(6, cfg.dim) * 0.1
    ctx = torch.randn(769, cfg.dim)
    ang = odit.rope_tables(128, f, h, w)
    nh, hd = cfg.num_heads, cfg.dim // cfg.num_heads
    q1, k1, v1 = (torch.randn(L1, nh, hd) for _ in range(3))
    hs = 4
    qL, kL, vL = torch.randn(L1, hs, hd), torch.randn(L, hs, hd), torch.randn(L, hs, hd)
    with torch.no_grad():
        t_blk = _median3(lambda: odit.block(tok, e0, ctx, W, 0, cfg, ang))
        t_core1 = _median3(lambda: odit.attention(q1, k1, v1))
        t_coreL = _median3(lambda: odit.attention(qL, kL, vL))
    del W
    t_block_true = (t_blk - t_core1) * (L / L1) + t_coreL * (nh / hs) * (L / L1)
    Wv = ovae.random_weights(seed=2)
    Fs, Hs, Ws = 5, 96, 96
    xs = torch.rand(1, 3, Fs, Hs, Ws) * 2 - 1
    with torch.no_grad():
        t_vae = _median3(lambda: ovae.decode(Wv, ovae.encode_mode(Wv, xs)))
    t_vae_true = t_vae * (frames * height * width) / (Fs * Hs * Ws)
    return dict(cores=cores, t_dit_forward_s=40 * t_block_true, t_vae_roundtrip_s=t_vae_true,
                sample=f"oracle fp32, {cores} threads, median of 3 after warm-up: DiT block (d=5120, 40 heads, FFN 13824) at L1={L1}: {t_blk:.2f}s "
                       f"(its self-attention core {t_core1:.2f}s); core with the true key length {L1} q x {L} k on {hs}/40 heads: {t_coreL:.2f}s; "
                       f"VAE encode+decode {Fs}x{Hs}x{Ws}: {t_vae:.2f}s; block@L = (block - core)*L/L1 + core_L*(40/{hs})*L/L1 = "
                       f"{t_block_true:.1f}s, x40 blocks per forward; VAE scaled by pixel-frames to {frames}x{height}x{width}: {t_vae_true:.0f}s; "
                       "steps/s = steps / sum(count x unit time) over the timed step mix (extrapolation)")


def cpu_baseline_longcat():
    """Oracle (CPU port, fp32) timed on this host: one LongCat block at the released width on a bounded token sample + the VAE sample
    of cpu_baseline(); returns flop rates."""
    from oracle import longcat_dit as olc
    from oracle import vae as ovae

    torch.manual_seed(0)
    cfg = olc.LongCatConfig(depth=1)
    W = olc.random_weights(cfg, seed=1)
    T, h, w = 2, 32, 32  # 512 tokens
    Ls = T * (h // 2) * (w // 2)
    x, cap = torch.randn(16, T, h, w), torch.randn(64, cfg.caption_channels)
    with torch.no_grad():
        t0 = time.time()
        olc.forward(W, cfg, x, torch.tensor([0.0, 500.0]), cap, None, num_cond_latents=1)
        t_blk = time.time() - t0
    C, Hd = cfg.hidden_size, cfg.ffn_hidden
    flop_blk = 2.0 * Ls * C * (6 * C + 3 * Hd) + 4.0 * Ls * Ls * C
    del W
    Wv = ovae.random_weights(seed=2)
    Fs, Hs, Ws = 5, 64, 64
    with torch.no_grad():
        t0 = time.time()
        ovae.decode(Wv, ovae.encode_mode(Wv, torch.rand(1, 3, Fs, Hs, Ws) * 2 - 1))
        t_vae = time.time() - t0
    return dict(cores=torch.get_num_threads(), dit_flops_per_s=flop_blk / t_blk, vae_flops_per_s=(5.19e6 + 8.70e6) * Fs * Hs * Ws / t_vae,
                sample=f"oracle fp32: 1 LongCat block (d=4096, 32 heads, SwiGLU 11008) + embeddings at L={Ls} tokens in {t_blk:.2f}s + VAE "
                       f"encode+decode of {Fs}x{Hs}x{Ws} in {t_vae:.2f}s; extrapolated by algorithmic FLOPs to the timed step mix")


def main_longcat(a):
    """`--workload longcat`: BASELINE config 4's model on the same contract -- LongCat-Video (13.6 B) guided i2v, 93 frames x 480 x 832,
    50-step schedule, IRR (3 rounds) + FLF + DSG + CFG-zero for the first 20 steps.  The timed window holds guided and plain steps in the
    job's 20 : 30 proportion."""
    rank, local_rank, world = rank_env(a)
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    from worldforge_amd import dit as wdit
    from worldforge_amd import parallel
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    # WF_FORCE_COMM=1: a one-rank process group, so that a one-GPU box runs the sharded code path over RCCL itself (debug / CI aid)
    comm = parallel.init(world, rank, local_rank) if (world > 1 or os.environ.get("WF_FORCE_COMM")) else None
    if a.as_rank_of > 1:   # one simulated rank of N on this GPU (parallel.LoopbackComm): the compute-bound ceiling of the N-GPU job
        if world != 1:
            raise SystemExit("bench.py: --as-rank-of is a one-process mode")
        comm = simulated_comm(a)
    cfg = LongCatConfig(depth=a.layers if a.layers != 40 else 48)
    frames = a.frames if a.frames != 81 else 93
    t0 = time.time()
    model = LongCatVideoTransformer3DModel(cfg, device, comm=comm).init_random(seed=0)
    vae = AutoencoderKLWan(device, comm=comm).init_random(seed=1)
    pipe = LongCatVideoPipeline(vae, FlowMatchEulerDiscreteScheduler(shift=12.0), model, device=device)
    g = torch.Generator().manual_seed(42)
    image = torch.rand(3, a.height, a.width, generator=g)
    ref = torch.rand(1, 3, frames, a.height, a.width, generator=g)
    mask = (torch.rand(1, 1, frames, a.height // 8, a.width // 8, generator=g) > 0.4).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    pe, ne = (torch.randn(2, 1, 1, 512, cfg.caption_channels, generator=g) * 0.5).bfloat16()
    pm, nm = torch.zeros(1, 512, dtype=torch.int64), torch.zeros(1, 512, dtype=torch.int64)
    pm[:, :180] = 1
    nm[:, :120] = 1
    torch.cuda.synchronize()
    t_setup = time.time() - t0
    exchange = None
    if comm is not None and comm.world > 1:
        # one evaluation on 4 real-width blocks per candidate: a single forward (distilled: no CFG) or the CFG batch (lock-step pair against
        # two own-first forwards); 1 condition latent frame as in the job
        Tl = (frames - 1) // 4 + 1
        nb = 1 if a.distill else 2
        xcal = torch.randn((nb, 16, Tl, a.height // 8, a.width // 8), device=device).to(torch.bfloat16)
        tcal = torch.tensor([[0.0] + [500.0] * (Tl - 1)] * nb)
        ccal = torch.cat([pe, ne])[:nb].to(device)
        mcal = torch.cat([pm, nm])[:nb]
        names = ["chunked2", "chunked4", "chunked1", "bcast", "gather"] if a.distill else ["lockstep", "chunked2", "chunked4", "chunked1", "bcast", "gather"]
        ctx = None
        if not a.distill and comm.world % 2 == 0:   # the CFG batch as two CFG groups x sequence shards (longcat_pipeline.cfg_split)
            sub = comm.split(2)
            ctx = {"world": comm, "sub": sub, "pipe": pipe}
            # the CFG groups come FIRST for LongCat (the default unless another candidate is >= 3 % faster on the node): as one rank of 8 they
            # are 16-18 % ahead of the lock-step pair in compute (profiles/r5_e_longcat_cfg_asrank8_*: 1.39-1.42 vs 1.20 steps/s), more
            # than any exposed exchange of a 4-rank group could cost
            names = (["cfg2+chunked2", "cfg2+chunked1", "cfg2+gather"] if sub.world > 1 else ["cfg2+gather"]) + names
        if a.exchange.startswith("cfg2+") and ctx is None:
            raise SystemExit("bench.py: --exchange cfg2+... needs the CFG batch (not --distill) and an even number of ranks")

        def run_cal(name):
            if name.startswith("cfg2+"):
                b = ctx["sub"].group_index
                v = model(xcal[b:b + 1], tcal[b:b + 1], ccal[b:b + 1], mcal[b:b + 1], num_cond_latents=1).contiguous()
                both = torch.empty((comm.world,) + tuple(v.shape), dtype=v.dtype, device=device)
                comm.all_gather(both, v)
            else:
                model(xcal, tcal, ccal, mcal, num_cond_latents=1)

        exchange = calibrate_exchange(model, comm, run_cal, names, "depth", device, a.exchange, ctx=ctx)
        if a.as_rank_of > 1 and a.exchange == "auto" and not a.emulate_comm:
            apply_exchange(model, names[0], ctx)
            exchange.update(selected=names[0], selection="default (simulated rank: the calibration shows each mode's compute cost only)")
        del xcal
    K, Wm = a.steps, a.warmup
    n_g = min(K, max(1, round(0.4 * K))) if K > 1 else 1
    guide = Wm + n_g
    n_sched, cfg_scale = (16, 1.0) if a.distill else (50, 4.0)
    if Wm + K > n_sched:
        raise SystemExit(f"--warmup + --steps must fit the {n_sched}-step schedule")
    marks = {}

    class _Stop(Exception):
        pass

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    def hook(i, phase):
        if phase == "start" and i == Wm:
            barrier()
            marks["t0"] = time.perf_counter()
            wdit.PROFILE_ATTN = []
            wdit.PROFILE_COMM = [] if comm is not None else None
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "end" and i == Wm + K - 1:
            barrier()
            marks["t1"] = time.perf_counter()
            raise _Stop

    try:
        pipe.generate_i2v(image=image, height=a.height, width=a.width, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                          negative_prompt_attention_mask=nm, num_frames=frames, num_inference_steps=n_sched, use_distill=a.distill,
                          guidance_scale=cfg_scale, generator=torch.manual_seed(42), output_type="latent", video_ref=ref, mask=mask,
                          guided=True, resample_steps=3,
                          guide_steps=guide, resample_round=guide, omega=1.8, omega_resample=1.0, use_pca_channel_selection=True,
                          static=True, step_hook=hook)
    except _Stop:
        pass
    el = torch.tensor([marks["t1"] - marks["t0"]], dtype=torch.float64, device=device)
    if comm is not None:
        comm.all_reduce_max(el)
    elapsed = el.item()
    prof = wdit.PROFILE_ATTN or []
    wdit.PROFILE_ATTN = None
    torch.cuda.synchronize()
    attn_ms = [s.elapsed_time(e) for s, e in prof]
    cprof = wdit.PROFILE_COMM or []
    wdit.PROFILE_COMM = None
    comm_ms = [wdit.comm_wait_ms(e) for e in cprof]
    T = (frames - 1) // 4 + 1
    tpf = (a.height // 16) * (a.width // 16)
    L = T * tpf
    per_rank = None
    if comm is not None:
        # every rank's own figures: its noise-token self-attention time and how long its compute stream stalled per layer waiting for
        # windows of the K / V^T exchange (own-first sweeps: only what has not arrived when the attention gets to it)
        mine = torch.tensor([sum(attn_ms) / max(len(attn_ms), 1), sum(comm_ms) / max(len(comm_ms), 1), float(len(comm_ms))],
                            dtype=torch.float64, device=device)
        allr = torch.empty((comm.world, 3), dtype=torch.float64, device=device)
        comm.all_gather(allr, mine)
        per_rank = [{"rank": r, "attn_avg_ms": v[0], "comm_exposed_ms_per_layer": v[1], "layers_timed": int(v[2])}
                    for r, v in enumerate(allr.cpu().tolist())]
    if rank == 0:
        gms = [1e3 * (marks[("e", i)] - marks[("s", i)]) for i in range(Wm, Wm + K) if i < guide]
        pms = [1e3 * (marks[("e", i)] - marks[("s", i)]) for i in range(Wm, Wm + K) if i >= guide]
        out = {"metric": "denoising steps/sec (93f x 480p, LongCat-Video 13.6B)", "value": K / elapsed, "unit": "steps/s", "n_gpus": world,
               "steps": K, "warmup": Wm, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"LongCat-Video i2v, {frames}f {a.height}x{a.width}, "
                                      + ("distilled 16-step schedule, IRR x3 + FLF + DSG, no CFG; timed " if a.distill else
                                         "50-step schedule, IRR x3 + FLF + DSG + CFG-zero 4; timed ")
                                      + f"steps {Wm}..{Wm + K - 1} = {len(gms)} guided + {len(pms)} plain",
                          "tokens": L, "dit_layers": cfg.depth, "dit_params_bytes": model.param_bytes(),
                          "parallelism": "single" if world == 1 else f"sp{world} (token-sharded DiT + row-sharded VAE, {_transport()})"},
               "window": {"guided": len(gms), "plain": len(pms), "guided_frac": len(gms) / max(K, 1)},
               "guided_step_ms": sum(gms) / len(gms) if gms else None, "plain_step_ms": sum(pms) / len(pms) if pms else None,
               "setup_s": t_setup}
        if a.as_rank_of > 1:
            out["metric"] += f" -- ONE simulated rank of {a.as_rank_of}: compute and local copies only, NOT a contract line"
            out["simulated_rank_of"], out["simulated_rank"] = a.as_rank_of, comm.rank
            out["config"]["parallelism"] = (f"rank {comm.rank} of sp{a.as_rank_of} on one GPU (parallel.LoopbackComm: collectives served from local "
                                            "data); value = what the N-GPU job would reach if communication were free")
            if a.emulate_comm:
                out["metric"] += "; communication EMULATED by a bandwidth model (stream-ordered delays), not measured"
                out["comm_model"] = dict(comm.model)
                out["config"]["parallelism"] += " -- under the bandwidth model of `comm_model`"
            per_rank = per_rank[comm.rank:comm.rank + 1] if per_rank else per_rank
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["exchange"] = exchange
        if gms and pms:
            if a.distill:
                out["job16_steps_per_s"] = 16.0 / ((6 * out["guided_step_ms"] + 10 * out["plain_step_ms"]) / 1e3)
            else:
                out["job50_steps_per_s"] = 50.0 / ((20 * out["guided_step_ms"] + 30 * out["plain_step_ms"]) / 1e3)
        if attn_ms and world == 1 and a.as_rank_of <= 1:
            avg = sum(attn_ms) / len(attn_ms)
            flop = 4.0 * (L - tpf) * L * 128 * cfg.num_heads
            ach = flop / (avg * 1e-3) / 1e12
            lc_kernel = "k_attn_w4<4> (pre-scaled Q)" if model.attn_prescale else "k_attn_w4<0>"
            out["roofline"] = {"kernel": lc_kernel + " (LongCat noise-token self-attention, attention.py:133-134)", "bound": "mfma",
                               "achieved": ach, "peak": MFMA_PEAK_TFLOPS_BF16, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS_BF16,
                               "traffic": None, "launches": len(attn_ms), "avg_launch_ms": avg, "flop_per_launch": flop}
        if cfg.depth != 48:
            out["invalid_reason"] = f"debug run with {cfg.depth} DiT blocks (the named model has 48)"
        if not a.no_cpu_baseline and world == 1 and a.as_rank_of <= 1:
            cb = cpu_baseline_longcat()
            C, Hd = cfg.hidden_size, cfg.ffn_hidden
            fwd = cfg.depth * (2.0 * L * C * (6 * C + 3 * Hd) + 4.0 * L * L * C)
            vae_flop = (5.19e6 + 8.70e6) * frames * a.height * a.width
            per = 1 if a.distill else 2  # DiT forwards per evaluation (CFG pair or not)
            t_cpu = (len(gms) * 3 * per + len(pms) * per) * fwd / cb["dit_flops_per_s"] + len(gms) * vae_flop / cb["vae_flops_per_s"]
            out["cpu_baseline"] = {"value": K / t_cpu, "unit": "steps/s", "cores": cb["cores"], "kind": "port", "sample": cb["sample"]}
        emit_json(out)
    if comm is not None:
        comm.barrier()
        shutdown_comm()


# ------------------------------------------------------------------------------------------------------------------------------------
# N > 1: which K / V^T exchange does THIS node hide best?  Timed on a few real-width layers before the timed window (VERDICT r4 #1d).
# ------------------------------------------------------------------------------------------------------------------------------------
EXCHANGES = {
    # name: (pair_lockstep, exchange_mode, exchange_chunks) -- worldforge_amd/parallel.py KVExchange, dit.WanTransformer3DModel attributes
    "lockstep": (True, "gather", 1),    # CFG pair one layer apart, ONE all-gather per layer hidden under the other branch (bit-identical to 1 GPU)
    "chunked2": (False, "chunked", 2),  # forwards one after the other; 2 all-gathers per layer, own shard first, then every peer's chunk g
    "chunked4": (False, "chunked", 4),
    "chunked1": (False, "chunked", 1),  # one all-gather, own shard first
    "bcast": (False, "bcast", 1),       # per-source broadcasts (K, V^T and bounds of a source in ONE collective), own shard first
    "gather": (False, "gather", 1),     # one all-gather, one launch, nothing overlapped but the Q projection
}


def apply_exchange(model, name: str, ctx=None):
    """Set the exchange candidate `name` on the model.  "cfg2+<mode>" (Wan with CFG, even world): the job's ranks as two CFG groups x
    sequence shards (SURVEY 8e "P = 8 = 2 x 4"; parallel.Comm.split, pipeline.cfg_split) -- ctx = dict(world=, sub=, pipe=) carries the two
    communicators and the pipeline whose CFG branch is switched."""
    split = name.startswith("cfg2+")
    model.pair_lockstep, model.exchange_mode, model.exchange_chunks = EXCHANGES[name[5:] if split else name]
    if ctx is not None:
        want = (ctx["sub"] if ctx["sub"].world > 1 else None) if split else ctx["world"]
        if model.comm is not want:
            model.comm = want
        if ctx.get("pipe") is not None:
            ctx["pipe"].cfg_split = (ctx["world"], ctx["sub"].group_index) if split else None


def calibrate_exchange(model, comm, run, names, depth_attr: str, device, forced: str = "auto", depths=(4, 12), reps: int = 3, ctx=None):
    """Time one evaluation (`run(name)`: a CFG pair or a single forward) of a model cut to 4 and to 12 real-width layers with every
    exchange candidate (max over ranks, min of `reps`) and extrapolate linearly to the full depth -- the per-forward fixed cost (embeddings,
    head, velocity gather) weighs ten times more in a 4-layer model than in the real one and differs between the candidates (two forwards
    per rank and evaluation in lock-step, one in the CFG-group split), so a single shallow timing mis-ranks them (measured: round 5, one
    rank of 8).  Keeps the fastest estimate -- the FIRST name is the default and stays unless another one is >= 3 % faster; the timings are
    all-reduced, so every rank takes the same decision.  -> dict for the JSON line."""
    if forced != "auto":
        apply_exchange(model, forced, ctx)
        return {"selected": forced, "selection": "forced by --exchange"}
    full = getattr(model.cfg, depth_attr)
    depths = sorted({min(full, d) for d in depths})
    timed = {name: {} for name in names}
    failed = {}
    try:
        for d in depths:
            setattr(model.cfg, depth_attr, d)
            for name in names:
                if name in failed:
                    continue
                apply_exchange(model, name, ctx)
                try:
                    run(name)  # allocates this mode's buffers
                except (AssertionError, ValueError, RuntimeError) as e:
                    # a host-side refusal (shape the plan cannot serve, C-ABI argument check) is the same on every rank: drop the candidate
                    # -- but never the default, and never silently
                    if name == names[0]:
                        raise
                    failed[name] = f"{type(e).__name__}: {e}"[:300]
                    print(f"bench.py: exchange candidate {name} dropped: {failed[name]}", file=sys.stderr)
                    continue
                best = None
                for _ in range(reps):
                    torch.cuda.synchronize()
                    comm.barrier()
                    t0 = time.perf_counter()
                    run(name)
                    torch.cuda.synchronize()
                    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                    comm.all_reduce_max(el)
                    best = el.item() if best is None else min(best, el.item())
                timed[name][d] = 1e3 * best
            model.__dict__.pop("_ctx_cache", None)   # (prompt-context K / V of the cut-down model)
    finally:
        setattr(model.cfg, depth_attr, full)
    lo, hi = depths[0], depths[-1]
    est = {n: (t[lo] + (t[hi] - t[lo]) / (hi - lo) * (full - lo) if hi > lo else t[lo]) for n, t in timed.items() if n not in failed}
    default = names[0]
    fastest = min(est, key=est.get)
    chosen = fastest if est[fastest] < 0.97 * est[default] else default
    apply_exchange(model, chosen, ctx)
    for k in [k for k in model._ws if isinstance(k[0], str) and k[0].startswith("kvx")]:  # the candidates' exchange buffers
        del model._ws[k]
    return {"selected": chosen, "estimated_ms_per_evaluation": est, "timed_ms": {n: {str(d): v for d, v in t.items()} for n, t in timed.items() if n not in failed},
            **({"dropped": failed} if failed else {}),
            "selection": f"fastest linear extrapolation from {lo} and {hi} real-width layers to {full}, max over ranks, min of {reps}; "
                         f"'{default}' unless another is >= 3 % faster"}


# ------------------------------------------------------------------------------------------------------------------------------------
# "also": short driver-timed windows of BASELINE configs 3 and 4 after the headline window
# ------------------------------------------------------------------------------------------------------------------------------------
def _attn_frac(wdit, flop_per_launch):
    prof = wdit.PROFILE_ATTN or []
    wdit.PROFILE_ATTN = None
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in prof]
    if not ms:
        return None, None
    avg = sum(ms) / len(ms)
    return flop_per_launch / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS_BF16, avg


def also_wan_720p(pipe, model, cfg, device, frames=81):
    """BASELINE config 3 (Wan2.1-I2V-14B-720P, 81 frames, full IRR + FLF + DSG, CFG 4) on the resident 14B model: steps 14, 15, 16 of the
    50-step schedule = 1 guided + 2 plain, no warm-up step (the first step also pays the 720p buffers' first touch)."""
    from worldforge_amd import dit as wdit
    H, W, guide = 720, 1280, 15
    image, ref, mask, text, neg, img_emb = synthetic_inputs(frames, H, W, device)
    marks = {}

    def hook(i, phase):
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "begin" and i == guide - 1:
            wdit.PROFILE_ATTN = []

    pipe(image=image, height=H, width=W, num_frames=frames, num_inference_steps=50, guidance_scale=4.0, generator=torch.manual_seed(42),
         prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img_emb, output_type="latent", video_ref=ref, mask=mask, guided=True,
         resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0, resample_round=guide, use_pca_channel_selection=True,
         static=True, start_step=guide - 1, max_steps=3, step_hook=hook)
    L = ((frames - 1) // 4 + 1) * (H // 16) * (W // 16)
    frac, avg = _attn_frac(wdit, 4.0 * L * L * 128 * cfg.num_heads)
    g = 1e3 * (marks[("e", guide - 1)] - marks[("b", guide - 1)])
    pl = [1e3 * (marks[("e", i)] - marks[("b", i)]) for i in (guide, guide + 1)]
    p = sum(pl) / len(pl)
    return {"workload": f"Wan2.1-I2V-14B-720P, {frames}f {H}x{W}, 50-step schedule, full IRR+FLF+DSG, CFG 4; timed steps 14..16 = 1 guided + 2 plain, no warm-up step",
            "tokens": L, "steps_per_s": 50.0 / ((15 * g + 35 * p) / 1e3), "steps_per_s_basis": "the 50-step job's 15 guided : 35 plain mix of the timed step times",
            "guided_step_ms": g, "plain_step_ms": p, "attn_frac": frac, "attn_avg_launch_ms": avg}


def also_longcat(device, height=480, width=832, frames=93):
    """BASELINE config 4 (LongCat-Video distilled 480p, 16 steps + the 720p refine pass) on a random-init 13.6 B model: steps 1..3 of the
    distilled 16-step schedule (1 guided step = 3 IRR rounds + FLF + DSG, 2 plain; no CFG) after one guided warm-up step, then steps 0 and 1
    of the 704 x 1280 refine pass (block-sparse self-attention at 98 560 tokens; step 1 reported)."""
    from worldforge_amd import dit as wdit
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    cfg = LongCatConfig()
    model = LongCatVideoTransformer3DModel(cfg, device).init_random(seed=0)
    vae = AutoencoderKLWan(device).init_random(seed=1)
    pipe = LongCatVideoPipeline(vae, FlowMatchEulerDiscreteScheduler(shift=12.0), model, device=device)
    g = torch.Generator().manual_seed(42)
    image = torch.rand(3, height, width, generator=g)
    ref = torch.rand(1, 3, frames, height, width, generator=g)
    mask = (torch.rand(1, 1, frames, height // 8, width // 8, generator=g) > 0.4).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    pe, ne = (torch.randn(2, 1, 1, 512, cfg.caption_channels, generator=g) * 0.5).bfloat16()
    pm, nm = torch.zeros(1, 512, dtype=torch.int64), torch.zeros(1, 512, dtype=torch.int64)
    pm[:, :180] = 1
    nm[:, :120] = 1
    marks = {}

    class _Stop(Exception):
        pass

    def hook(i, phase):
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "start" and i == 1:
            wdit.PROFILE_ATTN = []
        if phase == "end" and i == 3:
            raise _Stop

    try:   # steps 0 and 1 guided, 2 and 3 plain; step 0 is the warm-up (a cold pipeline's first guided step measured 1-1.5 s long)
        pipe.generate_i2v(image=image, height=height, width=width, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                          negative_prompt_attention_mask=nm, num_frames=frames, num_inference_steps=16, use_distill=True, guidance_scale=1.0,
                          generator=torch.manual_seed(42), output_type="latent", video_ref=ref, mask=mask, guided=True, resample_steps=3,
                          guide_steps=2, resample_round=2, omega=1.8, omega_resample=1.0, use_pca_channel_selection=True, static=True,
                          step_hook=hook)
    except _Stop:
        pass
    T = (frames - 1) // 4 + 1
    tpf = (height // 16) * (width // 16)
    L = T * tpf
    frac, avg = _attn_frac(wdit, 4.0 * (L - tpf) * L * 128 * cfg.num_heads)
    gms = 1e3 * (marks[("e", 1)] - marks[("s", 1)])
    pms = sum(1e3 * (marks[("e", i)] - marks[("s", i)]) for i in (2, 3)) / 2
    out = {"workload": f"LongCat-Video 13.6B distilled i2v, {frames}f {height}x{width}, 16-step schedule, IRR x3 + FLF + DSG, no CFG; timed steps 1..3 = "
                       "1 guided + 2 plain after one guided warm-up step",
           "tokens": L, "steps_per_s": 16.0 / ((6 * gms + 10 * pms) / 1e3), "steps_per_s_basis": "the 16-step job's 6 guided : 10 plain mix of the timed step times",
           "guided_step_ms": gms, "plain_step_ms": pms, "attn_frac": frac, "attn_avg_launch_ms": avg}
    # ---- the 720p refine pass (pipeline_longcat_video.py:1271-1511) on the same weights with block-sparse self-attention
    model._ws.clear()
    torch.cuda.empty_cache()
    model.enable_bsa()
    stage1 = (torch.rand(frames, height, width, 3, generator=g) * 255).to(torch.uint8)
    image2 = torch.rand(3, 704, 1280, generator=g)
    rm = {"t0": time.perf_counter()}

    def rhook(i, what):
        torch.cuda.synchronize()
        rm[(what[0], i)] = time.perf_counter()
        if what == "end" and i == 1:
            raise _Stop

    try:
        pipe.generate_refine(stage1_video=stage1, height=704, width=1280, prompt_embeds=pe, prompt_attention_mask=pm, image=image2,
                             num_cond_frames=1, num_inference_steps=50, generator=torch.manual_seed(1), t_thresh=0.5,
                             spatial_refine_only=True, step_hook=rhook)
    except _Stop:
        pass
    out["refine_720p"] = {"workload": "generate_refine 704x1280, 93 stage-1 frames -> 28 latent frames = 98 560 tokens, block-sparse self-attention "
                                      "(sparsity 0.875), no CFG, t_thresh 0.5; steps 0 and 1 timed, step 1 reported",
                          "prepare_s": rm[("s", 0)] - rm["t0"], "step_ms": 1e3 * (rm[("e", 1)] - rm[("s", 1)]),
                          "first_step_ms": 1e3 * (rm[("e", 0)] - rm[("s", 0)])}
    return out


def simulated_comm(a):
    """parallel.LoopbackComm for --as-rank-of N (+ the bandwidth model of --emulate-comm)."""
    from worldforge_amd import parallel
    model = None
    if a.emulate_comm:
        ag, link, lat = (float(x) for x in a.emulate_comm.split(","))
        model = {"allgather_gbps": ag, "link_gbps": link, "latency_us": lat}
    return parallel.LoopbackComm(a.as_rank_of, a.as_rank if a.as_rank >= 0 else a.as_rank_of // 2, model)


def launch_ranks(n: int, argv, script: str = None) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL rendezvous on 127.0.0.1) and
    wait for them.  This parent never touches the GPU (no torch.cuda call that initialises HIP, no libwf_hip.so) and never
    re-execs: the ranks are children, their stdout / stderr are inherited (rank 0 prints the JSON line), and the exit code is
    non-zero if any rank fails.  (The reference's own multi-GPU entry has the same shape: run_upscale.py:71-77 reads
    RANK / LOCAL_RANK from a launcher.)"""
    import socket
    import subprocess

    share = bool(os.environ.get("WF_SHARE_GPU"))
    have = visible_gpu_count()  # from the environment / sysfs: no torch.cuda call, nothing that could initialise HIP in the launcher
    if not share and have is not None and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible (WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo runs all ranks on one GPU "
              "as a debug configuration)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r if r > 0 else 1
                    for q in pending:  # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT touching the HIP runtime: the *_VISIBLE_DEVICES lists if set, else the KFD topology
    (a node with simd_count > 0 is a GPU).  None if neither source is readable (the pre-check is then skipped: a rank that finds no
    device fails non-zero and the launcher propagates it)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout of every rank (seen with RCCL 2.26 on the
    GPU box, flushed at exit, i.e. AFTER the line), and other libraries may chat there too: keep a private duplicate of fd 1 for the JSON
    line and point fd 1 at stderr for everything else, in every rank, before anything is initialised."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(out: dict):
    data = (json.dumps(out) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def shutdown_comm():
    """Tear the process group down before exit (RCCL otherwise warns about leaked resources; LoopbackComm has none)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def rank_env(a):
    """(rank, local_rank, world) from the launcher's environment; --gpus must agree with WORLD_SIZE."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher: it starts the ranks itself)")
    if os.environ.get("WF_SHARE_GPU"):  # debug: all ranks on one GPU (with WF_COMM_BACKEND=gloo) to exercise the N > 1 path
        local_rank = 0
    return rank, local_rank, world


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)  # default window: 3 guided + 7 plain = the 30 % mix of the 50-step job
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--layers", type=int, default=40, help="DiT depth (40 = Wan2.1-14B; smaller only for debugging -> flagged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flow-backend", default="farneback", choices=["tdiff", "farneback"],
                    help="FLF motion backend: farneback (default) = what the installed reference executes, as the GPU restatement of "
                         "cv2.calcOpticalFlowFarneback (parity with a real cv2 unpinned); tdiff = the branch the reference runs only "
                         "when `import cv2` fails (golden-pinned)")
    ap.add_argument("--vae-precision", default="fp16x3", choices=["fp16x3", "bf16x3", "fp32", "bf16"],
                    help="fp16x3 (default; 'fp32' names the same mode): fp32-CLASS VAE contractions standing in for the reference's fp32 VAE "
                         "(INFER:185-189) -- three-term split operands on the matrix cores, fp16 parts (weights power-of-two scaled: 3e-6 rel. L2 from fp32 end to end), 3x the VAE MFMA "
                         "work; bf16x3: the same split on bf16 parts (~2^-16 per product; the default of rounds 2-3, same cost); bf16: every "
                         "VAE operand rounded to bf16 (faster, 2^-9 per operand)")
    ap.add_argument("--distill", action="store_true",
                    help="with --workload longcat: the distilled 16-step schedule without CFG (BASELINE config 4's first half; the "
                         "cfg_step_lora is a weight fold and does not change the cost)")
    ap.add_argument("--as-rank-of", type=int, default=0, metavar="N",
                    help="NOT a contract line: run, on this one GPU, the work of ONE rank of an N-rank job (parallel.LoopbackComm: collectives "
                         "served from local data, values meaningless) and report that rank's step time -- the compute-bound ceiling of the "
                         "N-GPU throughput.  --as-rank picks the rank (default N // 2: halo rows on both sides)")
    ap.add_argument("--as-rank", type=int, default=-1)
    ap.add_argument("--emulate-comm", default=None, metavar="AG_GBPS,LINK_GBPS,LATENCY_US",
                    help="with --as-rank-of: a BANDWIDTH MODEL of the interconnect (NOT a measurement): every collective of the simulated rank "
                         "also queues a stream-ordered delay of latency + bytes / rate -- all-gathers at AG_GBPS per rank (received bytes), "
                         "per-source broadcasts at LINK_GBPS (one xGMI link: 153), e.g. 330,153,20.  The exchange calibration then SELECTS "
                         "under that model and the line reports what each mode would expose")
    ap.add_argument("--exchange", default="auto", choices=["auto"] + list(EXCHANGES) + ["cfg2+" + k for k in EXCHANGES if k != "lockstep"],
                    help="N > 1: how the sequence-parallel self-attention exchanges K / V^T (worldforge_amd/parallel.py KVExchange).  auto "
                         "(default): time a few real-width layers with every candidate on THIS node before the timed window and keep the "
                         "fastest (the line carries `exchange` with the timings)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short windows of BASELINE configs 3 (720p) and 4 (LongCat distilled + refine step) that the default "
                         "1-GPU run appends to the line as `also`")
    ap.add_argument("--workload", default="wan", choices=["wan", "longcat"],
                    help="wan = the BASELINE metric (default); longcat = the same contract on LongCat-Video 13.6B guided i2v (config 4's model)")
    a = ap.parse_args(argv)
    if a.vae_precision == "fp32":
        a.vae_precision = "fp16x3"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher.  Nothing above or in here initialises the GPU in this process.
        sys.exit(launch_ranks(a.gpus, argv))
    claim_stdout()
    if a.workload == "longcat":
        return main_longcat(a)

    rank, local_rank, world = rank_env(a)
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")

    from worldforge_amd import dit as wdit
    from worldforge_amd import parallel
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    # WF_FORCE_COMM=1: a one-rank process group, so that a one-GPU box runs the sharded code path over RCCL itself (debug / CI aid)
    comm = parallel.init(world, rank, local_rank) if (world > 1 or os.environ.get("WF_FORCE_COMM")) else None
    if a.as_rank_of > 1:
        if world != 1:
            raise SystemExit("bench.py: --as-rank-of is a one-process mode")
        comm = simulated_comm(a)

    cfg = wdit.DiTConfig.wan_i2v_14b()
    cfg.num_layers = a.layers
    t0 = time.time()
    model = wdit.WanTransformer3DModel(cfg, device, comm=comm).init_random(seed=0)
    vae = AutoencoderKLWan(device, comm=comm, precision=a.vae_precision).init_random(seed=1)  # high-resolution stages row-sharded over the ranks
    sch = UniPCMultistepScheduler(flow_shift=3.0, flow_backend=a.flow_backend)
    pipe = WanImageToVideoPipeline(model, vae, sch, device=device)
    image, ref, mask, text, neg, img_emb = synthetic_inputs(a.frames, a.height, a.width, device)
    torch.cuda.synchronize()
    t_setup = time.time() - t0
    exchange = None
    if comm is not None and comm.world > 1:
        # one CFG evaluation (PIPE:593-610) on 4 real-width layers per candidate: the lock-step pair, the own-first single forwards, and
        # -- even world sizes -- the two CFG groups x sequence shards (one forward per rank + the velocity all-gather over the job)
        xcal = torch.randn((cfg.in_dim, (a.frames - 1) // 4 + 1, a.height // 8, a.width // 8), device=device).to(torch.bfloat16)
        names = ["lockstep", "chunked2", "chunked4", "chunked1", "bcast", "gather"]
        ctx = None
        if comm.world % 2 == 0:
            sub = comm.split(2)
            ctx = {"world": comm, "sub": sub, "pipe": pipe}
            if sub.world > 1:
                names += ["cfg2+chunked2", "cfg2+chunked1", "cfg2+gather"]
            else:   # two GPUs: one CFG branch per GPU needs no K / V^T exchange at all -- the default there (first name)
                names = ["cfg2+gather"] + names

        def run_cal(name):
            if name.startswith("cfg2+"):
                v = model.forward_tokens(xcal, 500.0, (text if ctx["sub"].group_index == 0 else neg)[0], img_emb[0]).contiguous()
                both = torch.empty((comm.world,) + tuple(v.shape), dtype=v.dtype, device=device)
                comm.all_gather(both, v)
            else:
                model.forward_tokens_pair(xcal, 500.0, text[0], neg[0], img_emb[0])

        forced = a.exchange
        if forced.startswith("cfg2+") and ctx is None:
            raise SystemExit("bench.py: --exchange cfg2+... needs an even number of ranks")
        exchange = calibrate_exchange(model, comm, run_cal, names, "num_layers", device, forced, ctx=ctx)
        if a.as_rank_of > 1 and a.exchange == "auto" and not a.emulate_comm:   # communication is free on a simulated rank: the timings are the modes' COMPUTE cost
            apply_exchange(model, names[0], ctx)
            exchange.update(selected=names[0], selection="default (simulated rank: the calibration shows each mode's compute cost only)")
        del xcal
    K, Wm = a.steps, a.warmup
    n_g = max(1, round(0.3 * K)) if K > 1 else 1
    n_g = min(n_g, K)
    guide = 15
    start = max(0, guide - Wm - n_g)
    marks = {}

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    def hook(i, phase):
        idx = i - start
        if phase == "begin" and idx == Wm:
            torch.cuda.synchronize()
            marks["calib0"] = box_calib_tflops(device)   # outside the timed window, on a chip the warm-up steps have heated
            barrier()
            marks["t0"] = time.perf_counter()
            wdit.PROFILE_ATTN = []
            wdit.PROFILE_COMM = [] if comm is not None else None
        if phase == "begin":
            torch.cuda.synchronize()
            marks[("b", i)] = time.perf_counter()
        if phase == "end":
            torch.cuda.synchronize()
            marks[("e", i)] = time.perf_counter()
        if phase == "end" and idx == Wm + K - 1:
            barrier()
            marks["t1"] = time.perf_counter()
            marks["calib1"] = box_calib_tflops(device)

    gen = torch.manual_seed(42)
    pipe(image=image, height=a.height, width=a.width, num_frames=a.frames, num_inference_steps=50, guidance_scale=4.0,
         generator=gen, prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img_emb, output_type="latent",
         video_ref=ref, mask=mask, guided=True, resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0,
         resample_round=guide, use_pca_channel_selection=True, static=True, start_step=start, max_steps=Wm + K, step_hook=hook)
    elapsed = marks["t1"] - marks["t0"]
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if comm is not None:
        comm.all_reduce_max(el)
    elapsed = el.item()

    # per-kernel roofline figure: self-attention launches inside the timed region (HIP events on the launch stream)
    prof = wdit.PROFILE_ATTN or []
    wdit.PROFILE_ATTN = None
    torch.cuda.synchronize()
    attn_ms = [s.elapsed_time(e) for s, e in prof]
    cprof = wdit.PROFILE_COMM or []
    wdit.PROFILE_COMM = None
    comm_ms = [wdit.comm_wait_ms(e) for e in cprof]
    T = (a.frames - 1) // 4 + 1
    L = T * (a.height // 16) * (a.width // 16)
    Lq = model.local_tokens(L)
    attn_flop = 4.0 * Lq * L * 128 * cfg.num_heads
    per_rank = None
    if comm is not None:
        # every rank's own figures: self-attention rate of its token shard, and how long its compute stream stalled per layer waiting
        # for the K / V^T exchange (= the communication NOT hidden under the other CFG branch's layer)
        mine = torch.tensor([sum(attn_ms) / max(len(attn_ms), 1), attn_flop, sum(comm_ms) / max(len(comm_ms), 1), float(len(comm_ms))],
                            dtype=torch.float64, device=device)
        allr = torch.empty((comm.world, 4), dtype=torch.float64, device=device)
        comm.all_gather(allr, mine)
        per_rank = [{"rank": r, "attn_avg_ms": v[0], "attn_tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else None,
                     "comm_exposed_ms_per_layer": v[2], "layers_timed": int(v[3])} for r, v in enumerate(allr.cpu().tolist())]
    guided_ms = [1e3 * (marks[("e", i)] - marks[("b", i)]) for i in range(start + Wm, start + Wm + K) if i < guide]
    plain_ms = [1e3 * (marks[("e", i)] - marks[("b", i)]) for i in range(start + Wm, start + Wm + K) if i >= guide]

    if rank == 0:
        # the resolution class named in the metric / workload strings follows the run's own size (`--height 720` printed "480p")
        res = {(480, 832): "480p", (832, 480): "480p", (720, 1280): "720p", (1280, 720): "720p"}.get((a.height, a.width), f"{a.height}x{a.width}")
        out = {
            "metric": f"denoising steps/sec ({a.frames}f x {res}, Wan2.1-14B)",
            "value": K / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": f"Wan2.1-I2V-14B-{res.upper()}, {a.frames}f {a.height}x{a.width}, 50-step schedule, full IRR+FLF+DSG, CFG 4; "
                            f"timed steps {start + Wm}..{start + Wm + K - 1} = {len(guided_ms)} guided + {len(plain_ms)} plain",
                "tokens": L, "dit_layers": cfg.num_layers, "dit_params_bytes": model.param_bytes(),
                "parallelism": "single" if world == 1 else f"sp{world} (token-sharded DiT with K/V all-gather + row-sharded VAE with halo all-gather, {_transport()})",
                # the prompt-context K / V of the cross-attention are computed once per prompt, not once per forward (bit-identical; the
                # reference recomputes them, model.py:215-218): work removed from the timed region, stated here (VERDICT r4 #4a)
                "ctx_cache": os.environ.get("WF_CTX_CACHE", "1") != "0",
                "flow_backend": a.flow_backend,
                # the Farneback branch is what an installed reference executes, but its GPU statement is checked against the
                # in-repo restatement of OpenCV only (no cv2 in the image or the reference tree); the tdiff branch is golden-pinned
                "flow_backend_parity": "oracle-only (cv2 unpinned)" if a.flow_backend == "farneback" else "reference goldens (g4, g6)",
                "vae_precision": a.vae_precision + {"bf16x3": " (3-term split-bf16 operands, fp32 accumulate: ~2^-16 per product, not IEEE fp32)",
                                                     "fp16x3": " (3-term split-fp16 operands hi.hi + lo.hi + hi.lo, fp32 accumulate, weight operands stored power-of-two scaled so that their lo parts are normal fp16; "
                                                               "measured 2.6e-6 / 3.4e-6 rel. L2 of mu / decode from the fp32 goldens; fp32-class, not IEEE fp32)"}.get(a.vae_precision, ""),
            },
            # what the timed window holds (`value` is only comparable between lines with the same mix; the default K = 10 is the
            # 50-step job's own 15 : 35)
            "window": {"guided": len(guided_ms), "plain": len(plain_ms), "guided_frac": len(guided_ms) / max(K, 1)},
            "guided_step_ms": sum(guided_ms) / len(guided_ms) if guided_ms else None,
            "plain_step_ms": sum(plain_ms) / len(plain_ms) if plain_ms else None,
            "setup_s": t_setup,
        }
        if guided_ms and plain_ms:
            g, p = out["guided_step_ms"], out["plain_step_ms"]
            out["job50_steps_per_s"] = 50.0 / ((15 * g + 35 * p) / 1e3)
        if "calib0" in marks and "calib1" in marks:
            # DIAGNOSTIC, not the contract: `value` is what this box did; `value_normalised` is what a box sustaining the reference
            # bare-MFMA rate would have done, with the measured elasticity of the bench against that rate (BOX_CALIB_EXPONENT)
            cal = 0.5 * (marks["calib0"] + marks["calib1"])
            out["box_calib_tflops"] = {"before": marks["calib0"], "after": marks["calib1"], "mean": cal, "reference": BOX_CALIB_REFERENCE_TFLOPS,
                                       "kernel": "wf_calib_mfma: register-only v_mfma_f32_32x32x16_bf16 stream, N(0,1) operands, 256 x 4 waves"}
            headline = (a.frames, a.height, a.width, a.layers) == (81, 480, 832, 40) and world == 1 and a.as_rank_of <= 1 and attn_ms
            if headline:   # the fit is for this workload only (the attention reference time is its 32 760-token launch)
                avg_attn = sum(attn_ms) / len(attn_ms)
                out["value_normalised"] = (out["value"] * (BOX_CALIB_REFERENCE_TFLOPS / cal) ** BOX_CALIB_EXPONENT
                                           * (avg_attn / BOX_ATTN_REFERENCE_MS) ** BOX_ATTN_EXPONENT)
                out["box_calib_tflops"].update(exponent=BOX_CALIB_EXPONENT, attn_reference_ms=BOX_ATTN_REFERENCE_MS, attn_exponent=BOX_ATTN_EXPONENT)
        if a.as_rank_of > 1:  # one simulated rank: label it so that it cannot be mistaken for a measurement of N GPUs
            out["metric"] += f" -- ONE simulated rank of {a.as_rank_of}: compute and local copies only, NOT a contract line"
            out["simulated_rank_of"], out["simulated_rank"] = a.as_rank_of, comm.rank
            out["config"]["parallelism"] = (f"rank {comm.rank} of sp{a.as_rank_of} on one GPU (parallel.LoopbackComm: collectives served from local "
                                            "data); value = what the N-GPU job would reach if communication were free")
            if a.emulate_comm:
                out["metric"] += "; communication EMULATED by a bandwidth model (stream-ordered delays), not measured"
                out["comm_model"] = dict(comm.model, note="every collective = local copies + a delay of latency_us + bytes / rate on its stream: all-gathers at "
                                                          "allgather_gbps per rank (received bytes), per-source broadcasts at link_gbps")
                out["config"]["parallelism"] += " -- under the bandwidth model of `comm_model`"
            per_rank = per_rank[comm.rank:comm.rank + 1] if per_rank else per_rank
        if per_rank is not None:
            out["per_rank"] = per_rank
        if exchange is not None:
            out["exchange"] = exchange
            if exchange.get("selected", "").startswith("cfg2+") and a.as_rank_of <= 1:
                out["config"]["parallelism"] = (f"cfg2 x sp{world // 2} (two CFG groups of {world // 2} rank(s): token-sharded DiT inside a group, one velocity all-gather "
                                                f"per evaluation between them; VAE row-sharded over all {world} ranks, {_transport()})")
        if attn_ms:
            avg = sum(attn_ms) / len(attn_ms)
            ach = attn_flop / (avg * 1e-3) / 1e12
            kern = "k_attn_w4<4>" if model.attn_prescale else "k_attn_w4<0>"
            body = "" if kern != "k_attn_w4<4>" else (", max-tracking body" if model.attn_track_max else ", un-tracked body (selected by the per-head norm bounds)")
            out["roofline"] = {"kernel": kern + (" (pre-scaled Q" + body + ")" if kern == "k_attn_w4<4>" else "") + " (DiT self-attention, model.py:149-154)",
                               "bound": "mfma", "achieved": ach,
                               "peak": MFMA_PEAK_TFLOPS_BF16, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS_BF16,
                               **attn_pmc_fields(kern, L, cfg.num_heads, world, a.as_rank_of > 1),
                               "launches": len(attn_ms), "avg_launch_ms": avg,
                               "flop_per_launch": attn_flop}
            if kern == "k_attn_w4<4>" and world == 1 and a.as_rank_of <= 1:
                tms = tracked_body_ms(model, L, cfg.num_heads)
                if tms:
                    out["roofline"]["tracked_body_avg_launch_ms"] = tms
                    out["roofline"]["tracked_body_frac"] = attn_flop / (tms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS_BF16
        if world == 1 and a.as_rank_of <= 1:
            out["flf_gate_ms"] = flf_gate_ms(sch, (1, 16, T, a.height // 8, a.width // 8), device)
        if per_rank is not None and plain_ms:
            # one DiT layer of one forward on this rank ~ plain step / (2 forwards x layers); the K / V^T exchange is meant to hide under
            # the other CFG branch's layer: more than 10 % of a layer exposed means the overlap is NOT working on this node
            nfw = 1.0 if (exchange or {}).get("selected", "").startswith("cfg2+") else 2.0   # forwards a rank runs per CFG evaluation
            layer_ms = out["plain_step_ms"] / (nfw * cfg.num_layers)
            worst = max(r["comm_exposed_ms_per_layer"] for r in per_rank)
            out["comm_exposed_frac_of_layer"] = worst / layer_ms
            if worst > 0.10 * layer_ms and a.as_rank_of <= 1:
                out["comm_exposed_over_budget"] = True
                print(f"bench.py: WARNING: K / V^T exchange exposed {worst:.2f} ms per layer = {100 * worst / layer_ms:.0f} % of a "
                      f"{layer_ms:.2f} ms layer (budget 10 %): communication is NOT hidden under compute on this node -- run tools/comm_probe.py",
                      file=sys.stderr)
        if a.layers != 40:
            out["invalid_reason"] = f"debug run with {a.layers} DiT layers (the named model has 40)"
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(L, a.frames, a.height, a.width)
            ng, npl = len(guided_ms), len(plain_ms)
            t_cpu = (ng * 4 + npl * 2) * cb["t_dit_forward_s"] + ng * 2 * cb["t_vae_roundtrip_s"]
            out["cpu_baseline"] = {"value": K / t_cpu, "unit": "steps/s", "cores": cb["cores"], "kind": "port", "sample": cb["sample"]}
        default_job = (a.frames, a.height, a.width, a.layers) == (81, 480, 832, 40)
        if world == 1 and a.as_rank_of <= 1 and not a.no_also and default_job:
            # BASELINE configs 3 and 4 on the driver's own box (headline keys above are final; these are short extra windows)
            also = []
            model._ws.clear()
            torch.cuda.empty_cache()
            t_also = time.time()
            also.append(also_wan_720p(pipe, model, cfg, device))
            del pipe, model, vae
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also.append(also_longcat(device))
            out["also"] = also
            out["also_s"] = time.time() - t_also
        emit_json(out)
    if comm is not None:
        comm.barrier()
        shutdown_comm()


if __name__ == "__main__":
    main()
