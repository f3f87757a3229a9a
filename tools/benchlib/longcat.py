"""`bench.py --workload longcat`: the same contract on LongCat-Video 13.6B guided i2v (BASELINE config 4's model)."""
from __future__ import annotations

import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from .cpu import cpu_baseline_longcat
from .exchange import apply_exchange, calibrate_exchange, simulated_comm
from .also import LONGCAT_VAE_NOTE
from .launcher import _transport, emit_json, progress, rank_env, shutdown_comm
from .measure import MFMA_PEAK_TFLOPS_BF16


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
    progress(f"rank processes up; initialising the process group ({world} rank(s))")
    comm = parallel.init(world, rank, local_rank) if (world > 1 or os.environ.get("WF_FORCE_COMM")) else None
    if a.as_rank_of > 1:   # one simulated rank of N on this GPU (parallel.LoopbackComm): the compute-bound ceiling of the N-GPU job
        if world != 1:
            raise SystemExit("bench.py: --as-rank-of is a one-process mode")
        comm = simulated_comm(a)
    groups = None
    if comm is not None:   # every process group of the job, created and exercised here (parallel.Comm.prepare)
        comm.halo_whole_job = bool(a.conservative)
        groups = comm.prepare(cfg_groups=0 if (a.conservative or a.distill or comm.world % 2) else 2,
                              halo_distances=() if a.conservative else parallel.halo_distances(comm.world))
        progress(f"process groups ready: {len(groups)}")
    cfg = LongCatConfig(depth=a.layers if a.layers != 40 else 48)
    frames = a.frames if a.frames != 81 else 93
    t0 = time.time()
    model = LongCatVideoTransformer3DModel(cfg, device, comm=comm).init_random(seed=0)
    # the VAE in the dtype the reference entry loads it (run_longcat_worldforge_single.py:205: torch_dtype=torch.bfloat16) unless told otherwise
    vae_dtype = torch.bfloat16 if a.vae_precision == "bf16" else torch.float32
    vae = AutoencoderKLWan(device, comm=comm, precision=a.vae_precision, dtype=vae_dtype).init_random(seed=1)
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
        if not a.distill and comm.world % 2 == 0 and not a.conservative:   # the CFG batch as two CFG groups x sequence shards (longcat_pipeline.cfg_split)
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
        if phase == "start" and i < Wm:
            progress(f"warm-up step {i + 1} of {Wm}")
        if phase == "start" and i == Wm:
            barrier()
            progress(f"timed window starts: {K} steps")
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
                          "parallelism": "single" if world == 1 else f"sp{world} (token-sharded DiT + row-sharded VAE, {_transport()})",
                          "vae_precision": LONGCAT_VAE_NOTE if a.vae_precision == "bf16" else a.vae_precision + " (fp32 module)"},
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
        if comm is not None and a.as_rank_of <= 1:
            out["rccl"] = parallel.rccl_info()
            out["collectives_used"] = sorted(comm.used)
            out["process_groups"] = {"count": len(groups), "kinds": sorted({k for k, _ in groups}), "conservative": bool(a.conservative)}
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
