"""GPU: the own-first sweeps over the packed K / V^T exchange buffers (parallel.KVExchange + dit.attention_exchange: wf_attn_fwd_part with
its second window, wf_attn_merge, seg_stride_bytes / kmax_stride) against the ONE-launch sweep over the same buffers, for every rank's
window order, at a tight tolerance (the real-collective check alone allowed 2e-2).  No communicator: every slot is filled
directly, so what is compared is exactly the kernel-side arithmetic of the modes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF = torch.bfloat16


class _Rank:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank


def _fill(ex, k_full, v_full, L):
    """k_full / v_full [H, L, 128] (keys in sequence order) -> every slot of every chunk buffer, as the producers + collectives leave them."""
    from worldforge_amd import _ffi, ops
    H = k_full.shape[0]
    for p in range(ex.P):
        lo = p * ex.S
        n_valid = max(0, min(ex.S, L - lo))
        for g in range(ex.G):
            r0 = 64 * ex.tile_bounds[g]
            r1 = max(r0, min(64 * ex.tile_bounds[g + 1], n_valid))
            kslot, vslot, kmslot = ex.k[g][p], ex.vt[g][p], ex.km[g][p]
            kslot.zero_()
            vslot.zero_()
            kmslot.zero_()
            if r1 > r0:
                kslot[:, :r1 - r0].copy_(k_full[:, lo + r0:lo + r1])
                vrows = v_full[:, lo + r0:lo + r1].permute(1, 0, 2).reshape(r1 - r0, H * 128).contiguous()   # [rows, H * 128]
                _ffi.call("wf_v_transpose", vrows.data_ptr(), H * 128, vslot.data_ptr(), r1 - r0, ex.chunk_len(g), H, ops.stream())
                kmslot.copy_(kslot.float().pow(2).sum(-1).max(-1).values)


@pytest.mark.parametrize("mode,chunks", [("chunked", 1), ("chunked", 2), ("chunked", 3), ("bcast", 1)])
@pytest.mark.parametrize("P,L,Lq,H", [(2, 1000, 300, 2), (4, 2000, 500, 2), (8, 4000, 260, 1), (3, 1500, 700, 2), (8, 8 * 4096 - 8, 512, 1)])
def test_own_first_sweeps_equal_the_one_launch_sweep(P, L, Lq, H, mode, chunks):
    from tests._tol import within
    from worldforge_amd import dit
    from worldforge_amd.parallel import KVExchange, shard_plan
    g = torch.Generator().manual_seed(P * 1000 + L)
    plan = shard_plan(L, P)
    scale = 1.4426950408889634 / math.sqrt(128.0)
    q = (torch.randn(H, Lq, 128, generator=g) * scale).to(BF).to(DEV)          # pre-scaled Q (softmax_scale = 0 form)
    k_full = torch.randn(H, L, 128, generator=g).to(BF).to(DEV)
    k_full[:, L // 3] *= 4.0                                                   # a spiked key: the partial maxima of the windows differ
    v_full = torch.randn(H, L, 128, generator=g).to(BF).to(DEV)
    qm = q.float().pow(2).sum(-1).max(-1).values.contiguous()
    # reference: ONE launch over the gathered single-buffer form (mode "gather" on the same data)
    ex0 = KVExchange(_Rank(P, 0), H, plan.shard_len, "gather", 1, DEV)
    _fill(ex0, k_full, v_full, L)
    ref = torch.empty(Lq, H * 128, dtype=BF, device=DEV)
    dit.attention(q, ex0.k[0], ex0.vt[0], ref, L, 0.0, nsplit=1, kmax2=ex0.km[0], qmax2=qm)
    # ... which itself equals the plain softmax (a loose sanity bar; the tight comparisons are between the kernel's own forms)
    want = torch.softmax(q.float() @ k_full.float().transpose(1, 2) * math.log(2.0), -1) @ v_full.float()
    assert (ref.float().view(Lq, H, 128).permute(1, 0, 2) - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    worst = 0.0
    for r in range(P):
        if plan.local_tokens(r) <= 0:
            continue
        ex = KVExchange(_Rank(P, r), H, plan.shard_len, mode, chunks, DEV)
        _fill(ex, k_full, v_full, L)
        for use_bounds in (True, False):   # un-tracked body (bounds given) and the max-tracking body
            out = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
            dit.attention_exchange(q, ex, out, L, 0.0, qm, use_bounds=use_bounds)
            assert torch.isfinite(out.float()).all(), (r, use_bounds)
            worst = max(worst, (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item())
        # a key PREFIX (LongCat's condition rows see the first keys only), through the same windows
        kv = max(64, (L // 5) // 8 * 8 + 3)
        refp = torch.empty(Lq, H * 128, dtype=BF, device=DEV)
        dit.attention(q, ex0.k[0], ex0.vt[0], refp, kv, 0.0, nsplit=1, kmax2=ex0.km[0], qmax2=qm)
        outp = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
        dit.attention_exchange(q, ex, outp, kv, 0.0, qm)
        worst = max(worst, (outp.float() - refp.float()).abs().max().item() / refp.float().abs().max().item())
    # the windows are merged exactly (flash combine) from fp32 partials: what differs from the one-launch sweep is the fp32 association
    # of the row sums in front of the ONE bf16 rounding of the output -- under two bf16 ulps of the largest output (2 x 2^-8 relative);
    # measured 3.9e-3 ... 6.8e-3 on an MI355X (profiles/r6_tolerances.txt)
    within(f"exchange.{mode}{chunks}.P{P}.L{L}", worst, 7.9e-3)


def test_every_sweep_releases_the_exchange_buffers(monkeypatch):
    """the next layer's producers write the rank's own slot while the communication stream may still send from it;
    every own-first sweep must end behind the LAST collective on EVERY rank -- also on the rank whose windows never needed the last
    source's event (rank P - 1 in "bcast": the last source is itself)."""
    from worldforge_amd import dit, parallel
    calls = []
    real = parallel.KVExchange.wait_all
    monkeypatch.setattr(parallel.KVExchange, "wait_all", lambda self: (calls.append((self.rank, self.mode)), real(self))[1])
    H, P, L, Lq = 1, 4, 1024, 128
    plan = parallel.shard_plan(L, P)
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(H, Lq, 128, generator=g) * 0.1).to(BF).to(DEV)
    k_full, v_full = torch.randn(H, L, 128, generator=g).to(BF).to(DEV), torch.randn(H, L, 128, generator=g).to(BF).to(DEV)
    qm = q.float().pow(2).sum(-1).max(-1).values.contiguous()
    for mode in ("chunked", "bcast"):
        for r in range(P):
            ex = parallel.KVExchange(_Rank(P, r), H, plan.shard_len, mode, 2, DEV)
            _fill(ex, k_full, v_full, L)
            steps, _ = parallel.sweep_plan(ex, L, 1)
            if mode == "bcast" and r == P - 1:
                assert all(st["wait"] is None or st["wait"] < P - 1 for st in steps)   # its windows never wait for its own broadcast
            out = torch.empty(Lq, H * 128, dtype=BF, device=DEV)
            dit.attention_exchange(q, ex, out, L, 0.0, qm)
    assert calls == [(r, m) for m in ("chunked", "bcast") for r in range(P)]
