"""N > 1: which K / V^T exchange does THIS node hide best?  Timed on a few real-width layers before the timed window (VERDICT r4 #1d)."""
from __future__ import annotations

import os
import sys
import time

import torch

from .launcher import progress

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
                progress(f"exchange calibration: {name} at {d} layers")
                err = None
                try:
                    run(name)  # allocates this mode's buffers
                except (AssertionError, ValueError) as e:
                    # a HOST-SIDE refusal (a shape the part-launch plan cannot serve), raised before the forward issued a collective.  Anything
                    # else (RuntimeError: HIP / RCCL / out of memory, possibly after some collectives went out) is not caught: the rank dies
                    # and the supervisor relaunches the job conservatively.
                    err = e
                # the plan depends on the rank: agree on the refusal over the job, so that every rank takes the same branch
                flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64, device=device)
                comm.all_reduce_max(flag)
                if flag.item() > 0:
                    if name == names[0]:   # never the default, and never silently
                        raise err if err is not None else RuntimeError(f"exchange candidate {name} was refused on another rank")
                    failed[name] = (f"{type(err).__name__}: {err}" if err is not None else "refused on another rank")[:300]
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

def simulated_comm(a):
    """parallel.LoopbackComm for --as-rank-of N (+ the bandwidth model of --emulate-comm)."""
    from worldforge_amd import parallel
    model = None
    if a.emulate_comm:
        ag, link, lat = (float(x) for x in a.emulate_comm.split(","))
        model = {"allgather_gbps": ag, "link_gbps": link, "latency_us": lat}
    return parallel.LoopbackComm(a.as_rank_of, a.as_rank if a.as_rank >= 0 else a.as_rank_of // 2, model)
