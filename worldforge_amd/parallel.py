"""Sequence-parallel plumbing for the DiT: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference's Wan path is single-GPU (multi-GPU is an open TODO, README.md:245); its only statement of how Wan scales
sequence length is the unused upstream USP code (wan/distributed/xdit_context_parallel.py:93-226: chunk the token dim
after patch-embed, per-rank RoPE slice, gather at the exit).  The native design here:

  * tokens are split into P contiguous shards of `shard_len` = ceil(L / P) rounded up to 64 (the attention KV tile), the last
    shard is short; everything in the DiT except self-attention is token-local (weights replicated: 28 GB of 288 GB);
  * per layer ONE all-gather of the K shard and ONE of the blocked V^T shard (each [H, shard_len, 128] bf16); the gathered
    tensor [P, H, shard_len, 128] is consumed in place by the attention kernel (segment addressing, no re-layout);
    xGMI is point-to-point, so the all-gather's P-1 peer transfers run on distinct links;
  * the velocity shards are all-gathered once per forward; scheduler / injection math is replicated (it is tiny) and the
    CPU-generator noise is drawn identically on every rank (same seed), so no broadcast is needed inside the loop.
Only all-gather / broadcast / barrier are used (BASELINE.json north_star); the single max-reduce is bench.py's timing.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist


@dataclass
class ShardPlan:
    L: int          # total tokens
    P: int          # ranks
    shard_len: int  # padded tokens per rank (multiple of 64)

    def bounds(self, rank: int) -> Tuple[int, int]:
        lo = min(rank * self.shard_len, self.L)
        hi = min(lo + self.shard_len, self.L)
        return lo, hi

    def local_tokens(self, rank: int) -> int:
        lo, hi = self.bounds(rank)
        return hi - lo

    @property
    def padded_total(self) -> int:
        return self.P * self.shard_len


def shard_plan(L: int, P: int) -> ShardPlan:
    per = (L + P - 1) // P
    return ShardPlan(L=L, P=P, shard_len=(per + 63) // 64 * 64)


class Comm:
    """Thin wrapper over a torch.distributed process group (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, world: int, rank: int, group=None):
        self.world, self.rank, self.group = world, rank, group
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        """out [P, *inp.shape] <- inp from every rank."""
        assert out.shape[0] == self.world and tuple(out.shape[1:]) == tuple(inp.shape) and out.is_contiguous() and inp.is_contiguous()
        if dist.get_backend(self.group) == "gloo":
            if inp.is_cuda:
                # debug configuration only (several ranks sharing one GPU, WF_COMM_BACKEND=gloo): gloo's all_gather takes host
                # tensors, so stage through the host
                host = torch.empty(out.shape, dtype=out.dtype)
                dist.all_gather([host[i] for i in range(self.world)], inp.cpu(), group=self.group)
                out.copy_(host)
            else:
                dist.all_gather([out[i] for i in range(self.world)], inp, group=self.group)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)
        return out

    def all_gather_async(self, out: torch.Tensor, inp: torch.Tensor):
        """Launch the all-gather on the communication stream after the work queued so far on the current stream; returns an
        event the consumer stream must wait on.  Lets the Q projection overlap the K/V exchange."""
        if self.stream is None:
            self.all_gather(out, inp)
            return None
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.all_gather(out, inp)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        inp.record_stream(self.stream)
        out.record_stream(self.stream)
        return ev

    def broadcast(self, t: torch.Tensor, src: int = 0):
        dist.broadcast(t, src=src, group=self.group)
        return t

    def exchange_segments_async(self, out: torch.Tensor):
        """out [P, ...]: slot `rank` already holds this rank's shard (its producer wrote it on the current stream); every other slot is
        filled by a broadcast from its owner -- P broadcasts in SOURCE order (every rank must issue the same collective sequence) on the
        communication stream, one event per source.  Where all_gather_async hands back ONE event for the whole exchange, the consumer can
        here start on slot s as soon as event s has fired (stream order: events 0 .. s-1 have fired by then): the attention kernel walks
        its own shard with no wait at all and the segments that have arrived while the rest is still in flight (dit.attention_segmented)
        -- the overlap a forward without a second CFG branch has no other way to get.  Returns the list of P events (None on CPU)."""
        assert out.shape[0] == self.world and out.is_contiguous()
        gloo = dist.get_backend(self.group) == "gloo"

        def bcast(src):
            if gloo and out.is_cuda:  # debug configuration (ranks sharing one GPU over gloo): stage through the host
                host = out[src].cpu()
                dist.broadcast(host, src=src, group=self.group)
                if src != self.rank:
                    out[src].copy_(host)
            else:
                dist.broadcast(out[src], src=src, group=self.group)

        if self.stream is None:
            for src in range(self.world):
                bcast(src)
            return [None] * self.world
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        evs = []
        with torch.cuda.stream(self.stream):
            for src in range(self.world):
                bcast(src)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                evs.append(ev)
        out.record_stream(self.stream)
        return evs

    def barrier(self):
        dist.barrier(group=self.group)

    def all_reduce_max(self, t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t


class LoopbackComm:
    """ONE process standing in for rank `rank` of a `world`-rank job: every collective is served from the caller's own data (all_gather
    fills every slot with the local tensor, on the communication stream like the real one), so a one-GPU box runs exactly the kernels, shapes
    and launch sequence of one rank of an N-GPU job -- compute and local copies only, no interconnect; the VALUES are meaningless (every
    "peer" shard is a copy of this rank's).  `bench.py --as-rank-of N` times it: the per-rank step time is the compute-bound ceiling of the
    scaling curve, measured instead of guessed while no multi-GPU node is available."""

    def __init__(self, world: int, rank: int = 0):
        self.world, self.rank, self.group = world, rank, None
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        assert out.shape[0] == self.world and tuple(out.shape[1:]) == tuple(inp.shape)
        out.copy_(inp.unsqueeze(0).expand_as(out))
        return out

    all_gather_async = Comm.all_gather_async

    def exchange_segments_async(self, out: torch.Tensor):
        """Every peer slot is a copy of this rank's own (one copy + one event per source on the communication stream, like the real one)."""
        assert out.shape[0] == self.world and out.is_contiguous()
        if self.stream is None:
            for src in range(self.world):
                if src != self.rank:
                    out[src].copy_(out[self.rank])
            return [None] * self.world
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        evs = []
        with torch.cuda.stream(self.stream):
            for src in range(self.world):
                if src != self.rank:
                    out[src].copy_(out[self.rank])
                ev = torch.cuda.Event()
                ev.record(self.stream)
                evs.append(ev)
        out.record_stream(self.stream)
        return evs

    def broadcast(self, t: torch.Tensor, src: int = 0):
        return t

    def barrier(self):
        pass

    def all_reduce_max(self, t: torch.Tensor):
        return t


def init(world: int, rank: int, local_rank: int, backend: Optional[str] = None) -> Comm:
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = backend or os.environ.get("WF_COMM_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device(f"cuda:{local_rank}")
        dist.init_process_group(backend=backend, world_size=world, rank=rank, **kw)
    return Comm(world, rank)


def gather_rows(comm: Comm, local: torch.Tensor, plan: ShardPlan) -> torch.Tensor:
    """All-gather row shards [local_tokens, C] -> full [L, C] (pads to shard_len internally)."""
    C = local.shape[1]
    pad = torch.zeros((plan.shard_len, C), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    out = torch.empty((plan.P, plan.shard_len, C), dtype=local.dtype, device=local.device)
    comm.all_gather(out, pad)
    return out.view(plan.padded_total, C)[:plan.L]
