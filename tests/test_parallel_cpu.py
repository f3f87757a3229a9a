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
    # the segmented exchange (round 4): this rank's shard sits in its own slot, the peers' slots arrive by per-source broadcasts
    k_seg = torch.full((world, H, plan.shard_len, 8), -7.0)
    k_seg[rank] = k_loc
    evs = comm.exchange_segments_async(k_seg)
    ok &= len(evs) == world and all(e is None for e in evs)   # CPU: synchronous, nothing to wait for
    ok &= torch.equal(k_seg, k_all)
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
    """dit.segment_groups: the runs of consecutive physical key segments a rank walks -- its own shard first (no wait), then the peers in
    source order (the order of the per-source broadcasts), never crossing the own shard, at most peer_groups + 2 runs, every segment once."""
    from worldforge_amd.dit import segment_groups
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


def test_loopback_comm_serves_the_segmented_exchange():
    from worldforge_amd import parallel
    c = parallel.LoopbackComm(4, 2)
    out = torch.zeros(4, 3, 5)
    out[2] = torch.arange(15.0).view(3, 5)
    evs = c.exchange_segments_async(out)
    assert len(evs) == 4 and all(torch.equal(out[r], out[2]) for r in range(4))
