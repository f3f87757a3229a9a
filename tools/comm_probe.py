"""Diagnose the K / V^T exchange of the sequence-parallel DiT on a real N-GPU node (VERDICT r2 "Next" #3b).

    python tools/comm_probe.py --gpus 8                       # self-launching, one rank per GPU, RCCL
    WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo python tools/comm_probe.py --gpus 2 --iters 3     # plumbing check on a one-GPU box

The DiT hides each layer's all-gather of the K shard and of the blocked V^T shard ([H, shard, 128] bf16 each; 84 MB per rank and layer at
8 ranks of the 81 x 480 x 832 job) under the OTHER CFG branch's layer (dit.forward_tokens_pair).  Whether that works on a node is decided by
three things no one-GPU box can show, and this probe measures them at PRODUCTION sizes:

  A  the bare all-gather: time, bytes a rank receives per second, and the per-link rate if the transfers went all-pairs over the 7 xGMI
     links (a ring is per-link bound: SURVEY section 5);
  B  the self-attention launch of one rank's shard alone (k_attn_w4<4>, split KV sweep as dit.kv_splits decides);
  C  both at once (gather on the communication stream, attention on the compute stream): how much each slows the other -- RCCL's copy
     kernels need CUs, the attention kernel occupies every CU it runs on (160 KiB LDS, 512 registers per lane);
  D  what the DiT itself sees: `comm_exposed_ms_per_layer` of a few lock-step layers (HIP events around the compute stream's wait);
  E  the same for ONE forward on its own -- no second CFG branch to hide under: LongCat-Video, distilled schedules, guidance <= 1 -- in
     every exchange mode of parallel.KVExchange (round 5): "gather" (one all-gather of the packed [K | V^T | bounds] slots, one launch: only
     the Q projection overlaps), "chunked1/2/4" (G all-gathers, the attention walks its own shard at once and every peer's chunk g after
     all-gather g), "bcast" (per-source broadcasts, own shard first, peers in arrival order): dit.attention_exchange;
  F  the bare collectives of those modes on the packed buffers (all-gather, 2 / 4 chunk all-gathers, P broadcasts) against the two
     all-gathers of A.

`--sweep` (launcher only): the whole probe once per RCCL setting -- NCCL_MAX_NCHANNELS in {4, 8, 16, 32} and NCCL_PROTO in {default, Simple,
LL128} (the settings must be in the environment before the communicator exists, so every setting is a fresh set of rank processes) --
and a table of A / C / D / E per setting: how many CUs RCCL's copy kernels take from the one-workgroup-per-CU attention / GEMM kernels is
the thing no one-GPU box can show.

It also reports what RCCL decided (channels, algorithm / protocol lines of NCCL_DEBUG=INFO with the INIT and TUNING subsystems).
One JSON line on rank 0.  No data-path collective other than all-gather is used; the timing reduce is a max over ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _events(n):
    import torch
    return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def rccl_log_summary(path):
    """Channel count and algorithm / protocol lines from an NCCL_DEBUG=INFO log (best effort: formats differ between RCCL versions)."""
    out = {"channels": None, "tuning": [], "version": None}
    try:
        txt = open(path, errors="replace").read()
    except OSError:
        return out
    m = re.findall(r"Channel \d+/(\d+)", txt)
    if m:
        out["channels"] = max(int(x) for x in m)
    m = re.search(r"(RCCL version [^\n]+|NCCL version [^\n]+)", txt)
    if m:
        out["version"] = m.group(1).strip()
    seen = set()
    for line in txt.splitlines():
        if "AllGather" in line and ("Algo" in line or "algo" in line or "proto" in line):
            key = re.sub(r"^.*?NCCL INFO ", "", line).strip()
            if key not in seen and len(seen) < 12:
                seen.add(key)
                out["tuning"].append(key)
    return out


SWEEP = [{}] + [{"NCCL_MAX_NCHANNELS": str(c)} for c in (4, 8, 16, 32)] + [{"NCCL_PROTO": p} for p in ("Simple", "LL128")] \
    + [{"NCCL_MAX_NCHANNELS": "8", "NCCL_PROTO": "Simple"}]


def sweep(a, argv):
    """One full probe per RCCL setting (fresh rank processes each: the variables are read when the communicator is created)."""
    import subprocess
    rows = []
    for setting in SWEEP:
        env = dict(os.environ, **setting)
        for k in ("NCCL_MAX_NCHANNELS", "NCCL_PROTO", "NCCL_ALGO"):
            if k not in setting:
                env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, capture_output=True, text=True, timeout=3600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        name = " ".join(f"{k}={v}" for k, v in setting.items()) or "RCCL defaults"
        if r.returncode != 0 or not line:
            print(f"{name}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}", flush=True)
            continue
        d = json.loads(line[-1])
        rows.append((name, d))
        print(json.dumps({"setting": name, **d}), flush=True)
    f = lambda v, spec=".3f": "-" if v is None else format(v, spec)  # noqa: E731
    modes = ["gather1", "chunked1", "chunked2", "chunked4", "bcast1"]
    print("\n| RCCL setting | channels | A all-gather ms | A per-link GB/s | B attention ms | C attention under gather x | "
          "D pair: exposed ms / layer | " + " | ".join(f"F bare {m} ms" for m in modes) + " | "
          + " | ".join(f"E single {m}: layer ms (exposed)" for m in modes) + " |\n|" + "---|" * (7 + 2 * len(modes)))
    for name, d in rows:
        fb, es = d.get("F_bare_collectives_ms") or {}, d.get("E_single_forward") or {}
        print(f"| {name} | {(d.get('rccl') or {}).get('channels')} | {f(d['A_allgather_ms'])} | "
              f"{f(d['A_per_link_GBps_if_all_pairs'], '.1f')} | {f(d['B_attention_alone_ms'])} | {f(d['C_attention_slowdown'])} | "
              f"{f(d['D_comm_exposed_ms_per_layer'])} | " + " | ".join(f(fb.get(m)) for m in modes) + " | "
              + " | ".join(f"{f((es.get(m) or {}).get('layer_ms'))} ({f((es.get(m) or {}).get('exposed_ms_per_layer'))})" for m in modes) + " |")
    return 0 if rows else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tokens", type=int, default=32760, help="L of the job (81 x 480 x 832 -> 32760; 720p -> 75600)")
    ap.add_argument("--layers", type=int, default=4, help="DiT layers of part D (real width)")
    ap.add_argument("--sweep", action="store_true", help="launcher only: repeat the probe per RCCL channel count / protocol setting")
    a = ap.parse_args(argv)
    if a.sweep and "WORLD_SIZE" not in os.environ:
        sys.exit(sweep(a, [x for x in argv if x != "--sweep"]))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench
        logdir = tempfile.mkdtemp(prefix="wf_comm_probe_")
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
        os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(logdir, "rccl.%h.%p.log"))
        os.environ["WF_PROBE_LOGDIR"] = logdir
        sys.exit(bench.launch_ranks(a.gpus, argv, script=os.path.abspath(__file__)))

    import torch
    import bench
    bench.claim_stdout()
    rank, local_rank, world = bench.rank_env(a)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    from worldforge_amd import dit, parallel
    comm = parallel.init(world, rank, local_rank)
    H, L = 40, a.tokens
    plan = parallel.shard_plan(L, world)
    S = plan.shard_len
    lo, hi = plan.bounds(rank)
    Lq = hi - lo
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    bf = torch.bfloat16
    scale = 1.4426950408889634 / 128 ** 0.5
    kh = torch.zeros((H, S, 128), dtype=bf, device=dev)
    kh[:, :Lq] = torch.randn((H, Lq, 128), generator=g, device=dev).to(bf)
    vt = torch.randn((H, S // 64, 128, 64), generator=g, device=dev).to(bf)
    qh = (torch.randn((H, max(Lq, 1), 128), generator=g, device=dev) * scale).to(bf)
    k_all = torch.zeros((world, H, S, 128), dtype=bf, device=dev)
    v_all = torch.zeros((world, H, S // 64, 128, 64), dtype=bf, device=dev)
    ao = torch.empty((max(Lq, 1), H * 128), dtype=bf, device=dev)
    km = torch.stack([dit.head_max_norm2(kh, max(Lq, 1), torch.empty(H, device=dev)) for _ in range(world)])
    qm = dit.head_max_norm2(qh, max(Lq, 1), torch.empty(H, device=dev))
    nsplit = dit.kv_splits(H, max(Lq, 1), L)
    shard_bytes = kh.numel() * 2 + vt.numel() * 2

    def gather():
        e1 = comm.all_gather_async(k_all, kh)
        e2 = comm.all_gather_async(v_all, vt)
        return e1, e2

    def attend():
        dit.attention(qh, k_all, v_all, ao, L, 0.0, nsplit=nsplit, kmax2=km, qmax2=qm)

    def sync_all():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    cur = torch.cuda.current_stream()
    # warm-up (communicator set-up, first-touch)
    for _ in range(2):
        for ev in gather():
            if ev is not None:
                cur.wait_event(ev)
        attend()
    sync_all()

    # A: bare all-gather, timed on the communication stream
    ta = []
    for _ in range(a.iters):
        sync_all()
        t0 = time.perf_counter()
        evs = gather()
        for ev in evs:
            if ev is not None:
                ev.synchronize()
        torch.cuda.synchronize()
        ta.append(1e3 * (time.perf_counter() - t0))
    # B: attention alone
    tb = []
    for s, e in _events(a.iters):
        s.record()
        attend()
        e.record()
        tb.append((s, e))
    torch.cuda.synchronize()
    tb = [s.elapsed_time(e) for s, e in tb]
    # C: both at once
    tc_attn, tc_wall = [], []
    for _ in range(a.iters):
        sync_all()
        s, e = _events(1)[0]
        t0 = time.perf_counter()
        evs = gather()              # on the communication stream (it waits for the compute stream's queued work: none after the sync)
        s.record()
        attend()                    # on the compute stream, reading the PREVIOUS contents of k_all / v_all (a race on values, not on timing)
        e.record()
        for ev in evs:
            if ev is not None:
                ev.synchronize()
        torch.cuda.synchronize()
        tc_wall.append(1e3 * (time.perf_counter() - t0))
        tc_attn.append(s.elapsed_time(e))
    # D: a few real-width lock-step layers
    exposed = layer_ms = None
    MODES = [("gather", 1), ("chunked", 1), ("chunked", 2), ("chunked", 4), ("bcast", 1)]
    single = {m: (0.0, 0.0) for m in MODES}
    if a.layers > 0:
        cfg = dit.DiTConfig.wan_i2v_14b()
        cfg.num_layers = a.layers
        model = dit.WanTransformer3DModel(cfg, dev, comm=comm).init_random(seed=0)
        T = 21
        hw = {32760: (60, 104), 75600: (90, 160)}.get(L)
        if hw is not None:
            x = torch.randn((36, T, hw[0], hw[1]), generator=g, device=dev).to(bf)
            ctx_a = (torch.randn((200, 4096), generator=g, device=dev) * 0.1).to(bf)
            ctx_b = (torch.randn((120, 4096), generator=g, device=dev) * 0.1).to(bf)
            clip = torch.randn((257, 1280), generator=g, device=dev).to(bf)
            model.forward_tokens_pair(x, 500.0, ctx_a, ctx_b, clip)   # warm
            sync_all()
            dit.PROFILE_COMM = []
            t0 = time.perf_counter()
            model.forward_tokens_pair(x, 500.0, ctx_a, ctx_b, clip)
            torch.cuda.synchronize()
            pair_ms = 1e3 * (time.perf_counter() - t0)
            prof, dit.PROFILE_COMM = dit.PROFILE_COMM, None
            ex = [dit.comm_wait_ms(e) for e in prof]
            exposed = sum(ex) / max(len(ex), 1)
            layer_ms = pair_ms / (2 * a.layers)
            # E: ONE forward on its own in every exchange mode
            for mode in MODES:
                model.exchange_mode, model.exchange_chunks = mode
                model.forward_tokens(x, 500.0, ctx_a, clip)   # warm
                sync_all()
                dit.PROFILE_COMM = []
                t0 = time.perf_counter()
                model.forward_tokens(x, 500.0, ctx_a, clip)
                torch.cuda.synchronize()
                one_ms = 1e3 * (time.perf_counter() - t0)
                prof, dit.PROFILE_COMM = dit.PROFILE_COMM, None
                ex = [dit.comm_wait_ms(e) for e in prof]
                single[mode] = (sum(ex) / max(len(ex), 1), one_ms / a.layers)
    # F: the bare collectives of every mode on the packed exchange buffers
    tf = {}
    for mode in MODES:
        kvx = parallel.KVExchange(comm, H, S, mode[0], mode[1], dev)
        ts = []
        for _ in range(a.iters):
            sync_all()
            t0 = time.perf_counter()
            evs = kvx.launch()
            if evs[-1] is not None:
                evs[-1].synchronize()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        tf[mode] = _median(ts)
        del kvx
    nm = len(MODES)
    mine = torch.tensor([_median(ta), _median(tb), _median(tc_attn), _median(tc_wall), exposed or 0.0, layer_ms or 0.0]
                        + [single[m][0] for m in MODES] + [single[m][1] for m in MODES] + [tf[m] for m in MODES],
                        dtype=torch.float64, device=dev)
    allr = torch.empty((world, 6 + 3 * nm), dtype=torch.float64, device=dev)
    comm.all_gather(allr, mine)
    if rank == 0:
        r = allr.cpu()
        worst = r.max(dim=0).values.tolist()
        recv = (world - 1) * shard_bytes
        out = {"probe": "K / V^T all-gather of the sequence-parallel DiT", "n_gpus": world, "tokens": L, "shard_tokens": S,
               "backend": torch.distributed.get_backend(), "shard_bytes_k_plus_vt": shard_bytes, "bytes_received_per_rank": recv,
               "A_allgather_ms": worst[0], "A_recv_GBps_per_rank": recv / (worst[0] * 1e-3) / 1e9 if worst[0] > 0 else None,
               "A_per_link_GBps_if_all_pairs": shard_bytes / (worst[0] * 1e-3) / 1e9 if worst[0] > 0 and world > 1 else None,
               "xgmi_link_peak_GBps": 153.0,
               "B_attention_alone_ms": worst[1], "kv_splits": nsplit,
               "C_attention_under_allgather_ms": worst[2], "C_attention_slowdown": worst[2] / worst[1] if worst[1] > 0 else None,
               "C_both_wall_ms": worst[3], "C_overlap_efficiency": (worst[0] + worst[1]) / worst[3] if worst[3] > 0 else None,
               "D_layers": a.layers, "D_comm_exposed_ms_per_layer": worst[4] if exposed is not None else None,
               "D_layer_ms": worst[5] if layer_ms is not None else None,
               "D_exposed_frac_of_layer": (worst[4] / worst[5]) if exposed is not None and worst[5] > 0 else None,
               "E_single_forward": {f"{m}{c}": {"exposed_ms_per_layer": worst[6 + i] if exposed is not None else None,
                                                "layer_ms": worst[6 + nm + i] if exposed is not None else None} for i, (m, c) in enumerate(MODES)},
               "F_bare_collectives_ms": {f"{m}{c}": worst[6 + 2 * nm + i] for i, (m, c) in enumerate(MODES)},
               "per_rank": [{"rank": i, "allgather_ms": v[0], "attn_ms": v[1], "attn_under_gather_ms": v[2]} for i, v in enumerate(r.tolist())]}
        logdir = os.environ.get("WF_PROBE_LOGDIR")
        if logdir and os.path.isdir(logdir):
            logs = sorted(os.path.join(logdir, f) for f in os.listdir(logdir))
            out["rccl"] = rccl_log_summary(logs[0]) if logs else None
        ok = out["D_exposed_frac_of_layer"] is None or out["D_exposed_frac_of_layer"] <= 0.10
        out["verdict"] = "exchange hidden (<= 10 % of a layer exposed)" if ok else "EXCHANGE NOT HIDDEN: > 10 % of a layer exposed"
        bench.emit_json(out)
    comm.barrier()


if __name__ == "__main__":
    main()
