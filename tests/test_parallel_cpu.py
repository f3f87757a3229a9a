"""CPU (gloo, world size 2): the sequence-parallel plan, the all-gather layout the attention kernel consumes and the
velocity gather -- the N > 1 path of worldforge_amd/parallel.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from worldforge_amd.parallel import shard_plan


def test_shard_plan():
    p = shard_plan(32760, 8)
    assert p.shard_len == 4096 and p.padded_total == 32768
    assert [p.local_tokens(r) for r in range(8)] == [4096] * 7 + [4088]
    assert p.bounds(7) == (28672, 32760)
    p1 = shard_plan(32760, 1)
    assert p1.shard_len == 32768 and p1.local_tokens(0) == 32760
    p2 = shard_plan(4524, 4)
    assert p2.shard_len % 64 == 0 and sum(p2.local_tokens(r) for r in range(4)) == 4524
    for L, P in ((75600, 8), (4524, 2), (100, 3)):
        pl = shard_plan(L, P)
        assert sum(pl.local_tokens(r) for r in range(P)) == L and pl.shard_len * P >= L


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, L, H, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from worldforge_amd import parallel
    comm = parallel.init(world, rank, 0, backend="gloo")
    plan = parallel.shard_plan(L, world)
    lo, hi = plan.bounds(rank)
    g = torch.Generator().manual_seed(0)
    k_full = torch.randn(H, L, 8, generator=g)          # stands in for [H, L, 128]
    y_full = torch.randn(L, 4, generator=g)
    # per-rank K shard in the kernel's layout [H, shard_len, D], zero pad rows
    k_loc = torch.zeros(H, plan.shard_len, 8)
    k_loc[:, :hi - lo] = k_full[:, lo:hi]
    k_all = torch.empty(world, H, plan.shard_len, 8)
    ev = comm.all_gather_async(k_all, k_loc)
    assert ev is None  # CPU: synchronous
    # segment addressing of the attention kernel: key index g -> (seg = g // shard_len, row = g % shard_len)
    ok = True
    for gi in (0, 1, plan.shard_len - 1, min(plan.shard_len, L - 1), L - 1):
        seg, row = gi // plan.shard_len, gi % plan.shard_len
        ok &= torch.equal(k_all[seg, :, row], k_full[:, gi])
    ok &= float(k_all[world - 1, :, plan.local_tokens(world - 1):].abs().max()) == 0.0 if plan.local_tokens(world - 1) < plan.shard_len else True
    y = parallel.gather_rows(comm, y_full[lo:hi].contiguous(), plan)
    ok &= torch.equal(y, y_full)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    comm.all_reduce_max(t)
    ok &= t.item() == world - 1
    b = torch.full((3,), float(rank))
    comm.broadcast(b, src=0)
    ok &= float(b.sum()) == 0.0
    # the packed exchange (round 5, parallel.KVExchange): every rank writes its K shard, V^T shard and norm bounds into ITS slot of the
    # exchange buffers (chunk by chunk) and the collectives fill the peers' slots in place -- one collective per layer ("gather"), per
    # chunk ("chunked": chunk g = the g-th 1/G of every rank's keys) or per source ("bcast"), each carrying K AND V^T AND the bounds
    import torch.distributed as dist
    Hx, S = 2, plan.shard_len

    def fill(r, g, ex):  # what rank r puts into chunk g: a function of (rank, chunk) only
        sc = ex.chunk_len(g)
        base = 100.0 * r + 10.0 * g
        return (torch.full((Hx, sc, 128), base + 1), torch.full((Hx, sc // 64, 128, 64), base + 2), torch.full((Hx,), base + 3))

    for mode, chunks, want_calls in (("gather", 1, [("all_gather", None)]), ("chunked", 2, [("all_gather", None)] * 2),
                                     ("bcast", 1, [("broadcast", s) for s in range(world)])):
        ex = parallel.KVExchange(comm, Hx, S, mode, chunks, device="cpu")
        ok &= ex.collectives() == [(k, i if k == "all_gather" else s) for i, (k, s) in enumerate(want_calls)]
        for g in range(ex.G):
            kk, vv, mm = fill(rank, g, ex)
            ex.own_k(g).copy_(kk.bfloat16())
            ex.own_vt(g).copy_(vv.bfloat16())
            ex.own_km(g).copy_(mm)
        calls = []
        real_ag, real_bc = dist.all_gather, dist.broadcast
        dist.all_gather = lambda outs, inp, group=None: (calls.append(("all_gather", None, inp.numel())), real_ag(outs, inp, group=group))[1]
        dist.broadcast = lambda t, src=0, group=None: (calls.append(("broadcast", src, t.numel())), real_bc(t, src=src, group=group))[1]
        try:
            evs = ex.launch()
        finally:
            dist.all_gather, dist.broadcast = real_ag, real_bc
        # the issue order every rank must share, and ONE collective per chunk / source carrying the whole slot (K, V^T and bounds together)
        ok &= [(k, s_) for k, s_, _ in calls] == want_calls
        ok &= [n for _, _, n in calls] == [ex.bufs[g if mode == "chunked" else 0].shape[1] for g in range(len(calls))] if mode != "bcast" else \
            all(n == ex.bufs[0].shape[1] for _, _, n in calls)
        ok &= len(evs) == len(want_calls) and all(e is None for e in evs)   # CPU: synchronous, nothing to wait for
        for r in range(world):
            for g in range(ex.G):
                kk, vv, mm = fill(r, g, ex)
                ok &= torch.equal(ex.k[g][r].float(), kk) and torch.equal(ex.vt[g][r].float(), vv) and torch.equal(ex.km[g][r], mm)
        ex.wait_all()
    comm.barrier()
    ret[rank] = bool(ok)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_gather_layout():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 200, 3, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _split_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from worldforge_amd import parallel
    comm = parallel.init(world, rank, 0, backend="gloo")
    sub = comm.split(2)   # CFG groups x sequence shards (2 x world / 2)
    per = world // 2
    ok = (sub.world, sub.rank, sub.group_index, sub.ranks) == (per, rank % per, rank // per, list(range(rank // per * per, rank // per * per + per)))
    # collectives of a sub-group stay inside it; group-local source indices of the slot broadcasts map to the right global ranks
    mine = torch.full((3,), float(rank))
    out = torch.empty(per, 3)
    sub.all_gather(out, mine)
    ok &= out[:, 0].tolist() == [float(r) for r in sub.ranks]
    ex = parallel.KVExchange(sub, 2, 64, "bcast", 1, device="cpu")
    ex.own_k().fill_(float(rank))
    ex.launch()
    ok &= [float(ex.k[0][j].max()) for j in range(per)] == [float(r) for r in sub.ranks]
    # the row-sharded VAE's halo exchange: two-rank groups with each neighbour at distance d (d = ranks per row group), two phases
    for d in (1, 2):
        top, bottom = torch.full((2, 3), 10.0 * rank + 1), torch.full((2, 3), 10.0 * rank + 2)
        up, down = comm.neighbor_rows(top, bottom, d)
        ok &= (up is None) == (rank - d < 0) and (down is None) == (rank + d >= world)
        if up is not None:
            ok &= float(up.min()) == float(up.max()) == 10.0 * (rank - d) + 2      # the upper neighbour's BOTTOM row
        if down is not None:
            ok &= float(down.min()) == float(down.max()) == 10.0 * (rank + d) + 1  # the lower neighbour's TOP row
    # ... while the whole job still gathers over everyone (the velocity exchange between the two CFG groups, pipeline.cfg_split)
    allv = torch.empty(world, 3)
    comm.all_gather(allv, mine)
    ok &= allv[:, 0].tolist() == [float(r) for r in range(world)]
    comm.barrier()
    ret[rank] = bool(ok)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world4_cfg_groups():
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_split_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_kv_split_choice_depends_on_shapes_only():
    """dit.kv_splits: two KV splits only when a launch would leave a large part of the last round of workgroups idle
    (one rank's token shard at 8 ranks), never for the full-length launches; a pure function of the shapes."""
    from worldforge_amd.dit import kv_splits
    from worldforge_amd.parallel import shard_plan
    L = 32760
    expect = {1: 1, 2: 1, 4: 1, 8: 2}
    for P, want in expect.items():
        assert kv_splits(40, shard_plan(L, P).shard_len, L) == want
    assert kv_splits(40, 75600, 75600) == 1          # 720p, one GPU
    assert kv_splits(40, 4524, 4524) == 1            # short KV sweep: never split
    assert kv_splits(40, 32760, 512) == 1            # cross-attention


def test_loopback_comm_serves_collectives_locally():
    """parallel.LoopbackComm: one process standing in for one rank of N (bench.py --as-rank-of): all_gather fills every slot with the local
    tensor, the other collectives are no-ops; same call surface as Comm."""
    import torch
    from worldforge_amd import parallel
    c = parallel.LoopbackComm(4, 2)
    assert (c.world, c.rank) == (4, 2)
    x = torch.arange(6, dtype=torch.float32).view(2, 3)
    out = torch.empty(4, 2, 3)
    assert c.all_gather(out, x) is out and all(torch.equal(out[i], x) for i in range(4))
    out2 = torch.zeros(4, 2, 3)
    assert c.all_gather_async(out2, x + 1) is None and torch.equal(out2[3], x + 1)   # no stream on CPU: synchronous
    t = torch.tensor([3.0])
    assert c.all_reduce_max(t) is t and c.broadcast(t) is t
    c.barrier()
    for name in ("all_gather", "all_gather_async", "broadcast", "barrier", "all_reduce_max"):
        assert hasattr(parallel.Comm, name) and hasattr(c, name)


def test_segment_groups_own_shard_first_then_arrival_order():
    """parallel.segment_groups: the runs of consecutive physical key segments a rank walks -- its own shard first (no wait), then the peers in
    source order (the order of the per-source broadcasts), never crossing the own shard, at most peer_groups + 2 runs, every segment once."""
    from worldforge_amd.parallel import segment_groups
    assert segment_groups(8, 0) == [(0, 1), (1, 5), (5, 8)]
    assert segment_groups(8, 3) == [(3, 4), (0, 3), (4, 8)]
    assert segment_groups(8, 7) == [(7, 8), (0, 4), (4, 7)]
    assert segment_groups(2, 1) == [(1, 2), (0, 1)]
    assert segment_groups(4, 2, peer_groups=3) == [(2, 3), (0, 1), (1, 2), (3, 4)]
    for P in (2, 3, 4, 8):
        for r in range(P):
            for g in (1, 2, 3):
                runs = segment_groups(P, r, g)
                assert runs[0] == (r, r + 1) and len(runs) <= g + 2 and len(runs) <= 8
                segs = [s for a, b in runs for s in range(a, b)]
                assert sorted(segs) == list(range(P))
                assert all(not (a <= r < b) for a, b in runs[1:])


class _NoComm:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank


def _slots(n, k):
    tps = -(-n // k)
    return -(-n // tps)


@pytest.mark.parametrize("mode,chunks", [("chunked", 1), ("chunked", 2), ("chunked", 3), ("chunked", 4), ("bcast", 1)])
def test_sweep_plan_covers_every_valid_key_once_own_shard_first(mode, chunks):
    """parallel.sweep_plan (the part launches of an own-first sweep): for every rank, every valid 64-key tile of every chunk buffer is
    walked exactly once; the rank's own windows come first and wait for nothing; a peer window waits for the event of ITS chunk (chunked:
    event g = all-gather g) or of its LAST source (bcast) and reads the norm bounds of arrived slots only; the partial slots are
    contiguous, at most 12; second windows lie behind the first (the hole is the own segment).  Also for key prefixes (LongCat's
    condition rows see the first frames only)."""
    from worldforge_amd.parallel import MAX_ATTN_PARTS, KVExchange, shard_plan, sweep_plan
    for L, P in ((32760, 8), (32760, 4), (32760, 2), (4524, 3), (37440, 8), (1000, 8)):
        plan = shard_plan(L, P)
        for r in range(P):
            ex = KVExchange(_NoComm(P, r), 2, plan.shard_len, mode, chunks, device="cpu")
            assert sum(ex.chunk_len(g) for g in range(ex.G)) == plan.shard_len
            for kv_len in (L, L // 9 + 1, plan.shard_len, 64):
                for wgs in (640, 2560):
                    steps, nparts = sweep_plan(ex, kv_len, wgs)
                    assert 1 <= nparts <= MAX_ATTN_PARTS
                    seen = {g: [] for g in range(ex.G)}
                    slot, peers_started = 0, False
                    for st in steps:
                        g, (t0, t1, k1), w2 = st["chunk"], st["win"], st["win2"]
                        tps = ex.chunk_len(g) // 64
                        assert st["slot"] == slot and t0 < t1 and k1 >= 1
                        joined = t1 - t0
                        seen[g] += list(range(t0, t1))
                        if w2:
                            assert w2[0] >= t1 and w2[0] < w2[1] and (t1, w2[0]) == (r * tps, (r + 1) * tps)   # the hole is the own segment
                            joined += w2[1] - w2[0]      # round 6: both sides of the hole are ONE sequence for the same workgroups
                            seen[g] += list(range(w2[0], w2[1]))
                        assert st["nslots"] == _slots(joined, k1)
                        slot += st["nslots"]
                        # only the last launch may merge, and only when it is a single split with at least one slot in front of it
                        assert st["merge"] == (st is steps[-1] and len(steps) >= 2 and st["nslots"] == 1)
                        own = st["wait"] is None
                        if own:
                            assert not peers_started and t0 >= r * tps and t1 <= (r + 1) * tps and st["km"] == (r, 1)
                        else:
                            peers_started = True
                            a, n = st["km"]
                            segs = {t // tps for t in range(t0, t1)} | ({t // tps for t in range(w2[0], w2[1])} if w2 else set())
                            assert r not in segs and segs <= set(range(a, a + n))
                            if mode == "chunked":
                                assert st["wait"] == g and (a, n) == (0, P)
                            else:
                                assert st["wait"] == max(segs) and a + n - 1 == st["wait"]   # only slots that have arrived by then
                    assert slot == nparts
                    for g in range(ex.G):
                        valid = -(-ex.chunk_kv_len(kv_len, g) // 64)
                        assert sorted(seen[g]) == list(range(valid)), (L, P, r, g, kv_len)
                    assert sum(ex.chunk_kv_len(kv_len, g) for g in range(ex.G)) == min(kv_len, P * plan.shard_len)


def test_loopback_comm_serves_the_halo_exchange():
    from worldforge_amd import parallel
    top, bottom = torch.full((2, 3), 1.0), torch.full((2, 3), 2.0)
    for rank, want in ((0, (False, True)), (2, (True, True)), (3, (True, False))):
        up, down = parallel.LoopbackComm(4, rank).neighbor_rows(top, bottom, 1)
        assert ((up is not None), (down is not None)) == want
        assert up is None or torch.equal(up, bottom)
        assert down is None or torch.equal(down, top)
    up, down = parallel.LoopbackComm(8, 5).neighbor_rows(top, bottom, 2)
    assert up is not None and down is not None
    up, down = parallel.LoopbackComm(8, 6).neighbor_rows(top, bottom, 2)
    assert up is not None and down is None


def test_loopback_comm_serves_the_packed_exchange():
    from worldforge_amd import parallel
    c = parallel.LoopbackComm(4, 2)
    for mode, chunks in (("gather", 1), ("chunked", 2), ("bcast", 1)):
        ex = parallel.KVExchange(c, 2, 128, mode, chunks, device="cpu")
        for g in range(ex.G):
            ex.own_k(g).fill_(3.0 + g)
            ex.own_km(g).fill_(7.0)
        evs = ex.launch()
        assert len(evs) == (4 if mode == "bcast" else ex.G)
        for g in range(ex.G):
            assert all(torch.equal(ex.bufs[g][r], ex.bufs[g][2]) for r in range(4)) and float(ex.km[g].min()) == 7.0


def test_halo_pair_schedule_is_deadlock_free():
    """Comm.neighbor_rows runs the two-rank all-gathers in two phases: pair (lo, lo + d) in phase (lo // d) % 2.  For every chain length and
    neighbour distance: every adjacent pair (at distance d) is served exactly once, and within a phase no rank is in two pairs -- blocking
    two-rank collectives issued phase by phase can then never wait on each other in a cycle."""
    for P in range(1, 12):
        for d in range(1, 6):
            pairs = [(lo, lo + d) for lo in range(P - d)]
            by_phase = {0: [], 1: []}
            for lo, hi in pairs:
                by_phase[(lo // d) % 2].append((lo, hi))
            for ph, ps in by_phase.items():
                members = [r for p in ps for r in p]
                assert len(members) == len(set(members)), (P, d, ph, ps)
            # what each rank executes (the loop of Comm.neighbor_rows): its own view must be the same set of pairs
            mine = set()
            for r in range(P):
                for phase in (0, 1):
                    for lo in (r, r - d):
                        if lo >= 0 and lo + d < P and (lo // d) % 2 == phase:
                            mine.add((lo, lo + d))
            assert mine == set(pairs)
