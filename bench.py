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
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.also import also_longcat, also_wan_720p, also_wan_tf32_vae  # noqa: E402
from benchlib.cpu import cpu_baseline  # noqa: E402
from benchlib.exchange import EXCHANGES, apply_exchange, calibrate_exchange, simulated_comm  # noqa: E402,F401
from benchlib.launcher import (ATTEMPT_BUDGET_S, _transport, claim_stdout, emit_json, launch_ranks, progress, rank_env,  # noqa: E402,F401
                               shutdown_comm, supervise_under_launcher, visible_gpu_count)
from benchlib.longcat import main_longcat  # noqa: E402
from benchlib.measure import (BOX_ATTN_EXPONENT, BOX_ATTN_REFERENCE_MS, BOX_CALIB_EXPONENT, BOX_CALIB_REFERENCE_TFLOPS,  # noqa: E402
                              MFMA_PEAK_TFLOPS_BF16, attn_pmc_fields, box_calib_tflops, flf_gate_ms, synthetic_inputs, tracked_body_ms)


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
    ap.add_argument("--vae-precision", default=None, choices=["fp16x3", "bf16x3", "fp32", "fp16", "tf32", "bf16"],
                    help="default: what the workload's reference entry loads.  wan -> fp16x3 ('fp32' names the same mode): fp32-CLASS VAE "
                         "contractions standing in for the reference's fp32 VAE (INFER:185-189) -- three-term split fp16 operands, 3e-6 rel. L2 "
                         "from fp32 end to end, 3x the VAE MFMA work.  longcat -> bf16: a bf16 module with bf16 operands, the dtype "
                         "run_longcat_worldforge_single.py:205 loads.  fp16 ('tf32'): ONE fp16 term per operand = the multiplicand width of a "
                         "TF32 convolution (1e-3 from fp32); bf16x3: the three-term split on bf16 parts (2e-5)")
    ap.add_argument("--conservative", action="store_true",
                    help="N > 1: no process group beyond the job's own -- no CFG groups, the VAE's halo rows all-gathered over the whole job "
                         "-- what the supervisor's second attempt runs (with --exchange gather) after a failed or hung first one")
    ap.add_argument("--attempt-budget", type=float, default=ATTEMPT_BUDGET_S,
                    help="N > 1: wall seconds an attempt may take before the supervisor stops it and relaunches conservatively")
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
    if a.vae_precision is None:
        a.vae_precision = "bf16" if a.workload == "longcat" else "fp16x3"
    a.vae_precision = {"fp32": "fp16x3", "tf32": "fp16"}.get(a.vae_precision, a.vae_precision)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher.  Nothing above or in here initialises the GPU in this process.
        sys.exit(launch_ranks(a.gpus, argv, budget_s=a.attempt_budget))
    if a.gpus > 1 and not os.environ.get("WF_BENCH_CHILD"):
        # a rank started by torch.distributed.run: supervise the real rank as a child (one conservative relaunch on a failure / hang)
        sys.exit(supervise_under_launcher(argv, budget_s=a.attempt_budget))
    claim_stdout()
    if a.workload == "longcat":
        return main_longcat(a)

    rank, local_rank, world = rank_env(a)
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")

    from worldforge_amd import dit as wdit
    from worldforge_amd import ops as wops
    from worldforge_amd import parallel
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    # WF_FORCE_COMM=1: a one-rank process group, so that a one-GPU box runs the sharded code path over RCCL itself (debug / CI aid)
    progress(f"rank processes up; initialising the process group ({world} rank(s))")
    comm = parallel.init(world, rank, local_rank) if (world > 1 or os.environ.get("WF_FORCE_COMM")) else None
    if a.as_rank_of > 1:
        if world != 1:
            raise SystemExit("bench.py: --as-rank-of is a one-process mode")
        comm = simulated_comm(a)
    groups = None
    if comm is not None:
        # every process group of the job, created and exercised HERE (parallel.Comm.prepare): nothing creates a communicator later
        comm.halo_whole_job = bool(a.conservative)
        groups = comm.prepare(cfg_groups=0 if (a.conservative or comm.world % 2) else 2,
                              halo_distances=() if a.conservative else parallel.halo_distances(comm.world))
        progress(f"process groups ready: {len(groups)} ({', '.join(sorted({k for k, _ in groups}))})")

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
        if comm.world % 2 == 0 and not a.conservative:
            sub = comm.split(2)   # (created in comm.prepare above)
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
        if phase == "begin" and idx < Wm:
            progress(f"warm-up step {idx + 1} of {Wm} (schedule step {i})")
        if phase == "begin" and idx == Wm:
            torch.cuda.synchronize()
            marks["calib0"] = box_calib_tflops(device)   # outside the timed window, on a chip the warm-up steps have heated
            barrier()
            progress(f"timed window starts: {K} steps")
            marks["t0"] = time.perf_counter()
            wdit.PROFILE_ATTN = []
            wdit.PROFILE_COMM = [] if comm is not None else None
            wops.PROFILE_BLEND = []
        if phase == "begin":
            torch.cuda.synchronize()
            marks[("b", i)] = time.perf_counter()
        if phase == "end":
            torch.cuda.synchronize()
            marks[("e", i)] = time.perf_counter()
        if phase == "end" and idx == Wm + K - 1:
            barrier()
            marks["t1"] = time.perf_counter()
            progress(f"timed window ends: {marks['t1'] - marks['t0']:.1f} s")
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
    bprof = wops.PROFILE_BLEND or []
    wops.PROFILE_BLEND = None
    blend = [(s.elapsed_time(e) * 1e3, nb) for s, e, nb in bprof]
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
    crop_cols = vae._crop_range(vae.needed_columns(mask), a.width // 8) if vae.crop_to_mask else None   # (cached: no new read-back)
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
                # the two forwards of a CFG evaluation share what does not see the prompt (patch embedding + layer 0's self-attention block,
                # model.py:298-306): computed once per pair, bit-identical to two forwards -> 2 x 40 - 1 self-attention launches per pair
                "cfg_pair_shared_prefix": bool(model.pair_share_layer0),
                # the IRR injection decodes only the pixel columns its blend can see (mask != 1, + receptive-field halo; whole frames through
                # the latent-resolution stage): bit-identical blend (tests/test_gpu_vae.py, test_gpu_e2e.py); on THIS workload's mask:
                "vae_decode_needed_columns": {"enabled": bool(vae.crop_to_mask), "latent_columns_total": a.width // 8,
                                              "latent_columns_decoded": (list(crop_cols) if crop_cols else None)},
                "flow_backend": a.flow_backend,
                # the Farneback branch is what an installed reference executes, but its GPU statement is checked against the
                # in-repo restatement of OpenCV only (no cv2 in the image or the reference tree); the tdiff branch is golden-pinned
                "flow_backend_parity": "oracle-only (cv2 unpinned)" if a.flow_backend == "farneback" else "reference goldens (g4, g6)",
                "vae_precision": a.vae_precision + {"bf16x3": " (3-term split-bf16 operands, fp32 accumulate: ~2^-16 per product, not IEEE fp32)",
                                                     "fp16x3": " (3-term split-fp16 operands hi.hi + lo.hi + hi.lo, fp32 accumulate, weight operands stored power-of-two scaled so that their lo parts are normal fp16; "
                                                               "measured 2.6e-6 / 3.4e-6 rel. L2 of mu / decode from the fp32 goldens; fp32-class, not IEEE fp32)",
                                                     "fp16": " (ONE fp16 term per operand, fp32 accumulate: the multiplicand width of a TF32 convolution; 1.0e-3 / 1.4e-3 from the fp32 goldens -- NOT the headline configuration)"}.get(a.vae_precision, ""),
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
        if comm is not None and a.as_rank_of <= 1:
            # what the collectives travelled over and which ones were issued (north_star: broadcast / all-gather only on the data path; the
            # max-reduce is this file's timing and the exchange calibration)
            out["rccl"] = parallel.rccl_info()
            out["collectives_used"] = sorted(comm.used)
            out["process_groups"] = {"count": len(groups), "kinds": sorted({k for k, _ in groups}), "conservative": bool(a.conservative)}
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
        if blend:
            # SURVEY 8d "HBM GB/s for the injection kernels": the pixel blend of every IRR injection inside the timed window (HIP events on the
            # launch stream), algorithmic bytes = (3 + 1 + 3 + 3) x 4 B per pixel-frame (this rank's rows when the VAE is row-sharded)
            us, nb = sum(u for u, _ in blend) / len(blend), blend[0][1]
            out["hbm"] = {"kernel": "k_blend4_rgb (wf_blend_pixels, SCHED:1375-1380)", "bytes": nb, "avg_us": us, "launches": len(blend),
                          "achieved_GBps": nb / us / 1e3, "peak_GBps": 8000.0, "frac_of_8TBps": nb / us / 1e3 / 8000.0}
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
            also.append(also_wan_tf32_vae(pipe, device, out.get("guided_step_ms"), out.get("plain_step_ms")))
            model._ws.clear()
            torch.cuda.empty_cache()
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
