"""GPU: the kernels bench.py TIMES, checked against the oracle AT THE SIZES THEY ARE TIMED AT (VERDICT r2 "weak" #1).

bench.py's roofline kernel is `k_attn_w4<4>`: pre-scaled Q (softmax_scale = 0), body chosen per workgroup from the per-head norm bounds.
tests/test_gpu_fullsize.py / test_gpu_config3.py call `dit.attention(scale = 1/sqrt(128))`, i.e. `k_attn_w4<0>`.  Here the PRODUCTION
producer chain runs -- `wf_rmsnorm_heads(out_scale)` -> `wf_v_transpose` -> `wf_head_max_norm2` -> `wf_attn_fwd(scale = 0, kmax2, qmax2)`
(dit.py:579-586) -- on the C2 (32 760) and C3 (75 600) token grids, a debug counter (`wf_attn_debug_body_counter`) asserts WHICH body every
workgroup ran, and sampled rows are compared with oracle/dit.py (fp32 RMS-norm over 5120 channels, fp64 RoPE, fp32 softmax) computed from
the same bf16 projection output.  Also: the tracked body (no bounds; bounds too large), the split KV sweep at the 8-rank shard shape, one
real-width DiT layer at L = 32 760 and one released-width LongCat block at 37 440 tokens on sampled tokens, and the 98 560-token sparse
attention with the product's own selection.
"""
import math

import pytest
import torch

from oracle import bsa as obsa
from oracle import dit as odit
from oracle import longcat_dit as olc
from tests._tol import within

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF, F32 = torch.bfloat16, torch.float32
LOG2E = 1.4426950408889634
H, D_MODEL = 40, 5120
GRID_C2, GRID_C3 = (21, 30, 52), (21, 45, 80)


def _dev_randn(shape, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV, dtype=F32) * scale).to(dtype)


class _BodyCounter:
    """[tracked, un-tracked] workgroup counts of the pre-scaled-Q attention launches made inside the `with` block."""

    def __enter__(self):
        from worldforge_amd import _ffi
        self.c = torch.zeros(2, dtype=torch.int32, device=DEV)
        _ffi.call("wf_attn_debug_body_counter", self.c.data_ptr())
        return self

    def __exit__(self, *exc):
        from worldforge_amd import _ffi
        torch.cuda.synchronize()
        _ffi.call("wf_attn_debug_body_counter", None)
        self.tracked, self.untracked = (int(v) for v in self.c.cpu())


def _chain(grid, seed, q_gain=1.0):
    """The producers of dit.py:579-585 on a synthetic QKV-projection output.  -> dict of device tensors + the oracle's view of them."""
    from worldforge_amd import dit
    L = grid[0] * grid[1] * grid[2]
    Lp = (L + 63) // 64 * 64
    m = dit.WanTransformer3DModel(dit.DiTConfig.wan_i2v_14b(), DEV)
    qkv = _dev_randn((L, 3 * D_MODEL), seed)
    wq = (1 + 0.05 * _dev_randn((D_MODEL,), seed + 1, dtype=F32)) * q_gain
    wk = 1 + 0.05 * _dev_randn((D_MODEL,), seed + 2, dtype=F32)
    cos, sin = m._rope_tables(*grid)
    qh = torch.empty((H, L, 128), dtype=BF, device=DEV)
    kh = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    vt = torch.empty((H, Lp // 64, 128, 64), dtype=BF, device=DEV)
    m._heads(qkv, 0, wq, cos, sin, qh, L, out_scale=LOG2E / math.sqrt(128.0))
    m._heads(qkv, D_MODEL, wk, cos, sin, kh, L)
    m._vt(qkv, 2 * D_MODEL, vt, L)
    km = dit.head_max_norm2(kh, L, torch.empty(H, dtype=F32, device=DEV))
    qm = dit.head_max_norm2(qh, L, torch.empty(H, dtype=F32, device=DEV))
    return dict(L=L, Lp=Lp, grid=grid, qkv=qkv, wq=wq, wk=wk, qh=qh, kh=kh, vt=vt, km=km, qm=qm)


def _oracle_rows(c, rows, heads):
    """model.py:130-154 on the sampled rows / heads: RMS-norm over all 5120 channels (fp32), RoPE (fp64), softmax(q k^T / sqrt(128)) v."""
    L = c["L"]
    qkv = c["qkv"].float().cpu()
    ang = odit.rope_tables(128, *c["grid"])
    q = odit.rms_norm(qkv[rows, :D_MODEL], c["wq"].cpu(), 1e-6).view(len(rows), H, 128)[:, heads]
    k = odit.rms_norm(qkv[:, D_MODEL:2 * D_MODEL], c["wk"].cpu(), 1e-6).view(L, H, 128)[:, heads]
    v = qkv[:, 2 * D_MODEL:].reshape(L, H, 128)[:, heads]
    return odit.attention(odit.rope_apply(q, ang[rows]), odit.rope_apply(k, ang), v)   # [rows, heads, 128]


def _compare(out, want, rows, heads, tol_abs, tol_rel, name="attn"):
    got = torch.stack([out[rows, h * 128:(h + 1) * 128].float().cpu() for h in heads], dim=1)
    err = (got - want).abs().max().item()
    rel = (got - want).norm().item() / want.norm().item()
    within(name + ".max_abs", err, tol_abs)
    within(name + ".rel_l2", rel, tol_rel)
    return err, rel


def _wgs(Lq, nsplit=1):
    return ((H + 7) // 8) * 8 * (-(-Lq // 256)) * nsplit


ROWS = {GRID_C2: [0, 1, 31, 32, 255, 256, 4095, 4096, 16383, 20000, 32760 - 249, 32760 - 65, 32760 - 64, 32760 - 2, 32760 - 1],
        GRID_C3: [0, 1, 255, 256, 4096, 32759, 32760, 65535, 65536, 75519, 75520, 75600 - 81, 75600 - 16, 75600 - 2, 75600 - 1]}


@pytest.mark.parametrize("grid", [GRID_C2, GRID_C3], ids=["C2", "C3"])
def test_timed_self_attention_production_chain_vs_oracle(grid):
    """`k_attn_w4<4>` exactly as bench.py launches it: both bodies on the same operands, each asserted by the counter."""
    from worldforge_amd import dit
    c = _chain(grid, 900 + grid[1])
    L = c["L"]
    heads = (0, 23, 39)
    want = _oracle_rows(c, ROWS[grid], heads)
    # the bounds the kernel decides on: B^2 = max|q|^2 max|k|^2 (exp2 units) well under 2500 on unit-variance activations
    assert float((c["km"] * c["qm"]).max()) < 2500.0 and float(c["km"].min()) > 0
    out_u = torch.empty((L, H * 128), dtype=BF, device=DEV)
    with _BodyCounter() as n:
        dit.attention(c["qh"], c["kh"], c["vt"], out_u, L, 0.0, nsplit=1, kmax2=c["km"], qmax2=c["qm"])
    assert (n.tracked, n.untracked) == (0, _wgs(L)), (n.tracked, n.untracked)       # the body bench.py times
    # bars at <= 2x the error measured on an MI355X (profiles/r6_tolerances.txt, regenerated at HEAD every round-end run): max abs 2.5e-4 / 1.3e-4, rel L2 4.9e-3 / 5.2e-3 at C2 / C3
    tol_abs = 5e-4 if grid == GRID_C2 else 2.6e-4
    eu = _compare(out_u, want, ROWS[grid], heads, tol_abs, 1e-2, f"timed_attn.untracked.L{L}")
    out_t = torch.empty_like(out_u)
    with _BodyCounter() as n:
        dit.attention(c["qh"], c["kh"], c["vt"], out_t, L, 0.0, nsplit=1)           # no bounds -> running-max tracking (`model.attn_track_max = True`)
    assert (n.tracked, n.untracked) == (_wgs(L), 0), (n.tracked, n.untracked)
    et = _compare(out_t, want, ROWS[grid], heads, tol_abs, 1e-2, f"timed_attn.tracked.L{L}")
    # the two bodies differ only in the reference max m of each row (exact for any m up to fp32 rounding of exp2 / the row sums)
    within(f"timed_attn.bodies_agree.L{L}", (out_u.float() - out_t.float()).abs().max().item() / out_t.float().abs().max().item(), 2.0 ** -10)   # measured: identical (no rescale ever fires on unit-variance data); the bar allows a quarter of a bf16 ulp
    print(f"k_attn_w4<4> L={L}: un-tracked max abs {eu[0]:.2e} rel {eu[1]:.2e}; tracked {et[0]:.2e} rel {et[1]:.2e}")


def test_timed_self_attention_large_norms_select_the_tracked_body_c2():
    """q 6x larger: B^2 > 2500, so the kernel must take the tracked body BY DATA although bounds are given -- and stay correct on a softmax
    36x sharper in the exponent (scores ~ N(0, 36): a few keys dominate each row)."""
    from worldforge_amd import dit
    c = _chain(GRID_C2, 950, q_gain=6.0)
    L = c["L"]
    assert float((c["km"] * c["qm"]).min()) > 2500.0
    out = torch.empty((L, H * 128), dtype=BF, device=DEV)
    with _BodyCounter() as n:
        dit.attention(c["qh"], c["kh"], c["vt"], out, L, 0.0, nsplit=1, kmax2=c["km"], qmax2=c["qm"])
    assert (n.tracked, n.untracked) == (_wgs(L), 0), (n.tracked, n.untracked)
    want = _oracle_rows(c, ROWS[GRID_C2], (0, 17, 39))
    # sharp softmax: outputs are O(1) mixtures of a few values; the bf16 rounding of q / k moves a score by ~1e-2 -> ~1 % on p
    _compare(out, want, ROWS[GRID_C2], (0, 17, 39), 6e-2, 2.6e-2, "timed_attn.large_norms")   # measured 5.1e-2 / 1.3e-2


def test_nan_row_forces_the_tracked_body():
    """A NaN in K must not be hidden by fmaxf in the norm pass: the bound becomes +inf and the tracked body runs."""
    from worldforge_amd import dit
    L, Lp = 1000, 1024
    q = _dev_randn((H, L, 128), 1, 0.1)
    k = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    k[:, :L] = _dev_randn((H, L, 128), 2)
    k[3, 517, 5] = float("nan")
    vt = _dev_randn((H, Lp // 64, 128, 64), 3)
    km = dit.head_max_norm2(k, L, torch.empty(H, dtype=F32, device=DEV))
    qm = dit.head_max_norm2(q, L, torch.empty(H, dtype=F32, device=DEV))
    assert math.isinf(float(km[3])) and torch.isfinite(km[[0, 1, 2, 4]]).all()
    out = torch.empty((L, H * 128), dtype=BF, device=DEV)
    with _BodyCounter() as n:
        dit.attention(q, k, vt, out, L, 0.0, nsplit=1, kmax2=km, qmax2=qm)
    per_head = -(-L // 256)
    assert (n.tracked, n.untracked) == (per_head, (H - 1) * per_head), (n.tracked, n.untracked)
    assert torch.isnan(out[:, 3 * 128:4 * 128].float()).any() and torch.isfinite(out[:, :3 * 128].float()).all()


def test_timed_self_attention_split_kv_sweep_at_the_8_rank_shard_shape():
    """What one rank of 8 launches at C2 (dit.py:589-614): 4096 of its own query rows against the all-gathered K / V^T shards
    [8, H, 4096, 128] with per-shard norm bounds [8, H], KV sweep split in two (dit.kv_splits) + the merge kernel."""
    from worldforge_amd import dit
    c = _chain(GRID_C2, 960)
    L, P, S = c["L"], 8, 4096
    assert dit.kv_splits(H, S, L) == 2
    kpad = torch.zeros((H, P * S, 128), dtype=BF, device=DEV)
    kpad[:, :c["Lp"]] = c["kh"]
    k_all = kpad.view(H, P, S, 128).transpose(0, 1).contiguous()
    vpad = torch.zeros((H, P * S // 64, 128, 64), dtype=BF, device=DEV)
    vpad[:, :c["Lp"] // 64] = c["vt"]
    vt_all = vpad.view(H, P, S // 64, 128, 64).transpose(0, 1).contiguous()
    km_all = torch.stack([dit.head_max_norm2(k_all[p], min(S, L - p * S), torch.empty(H, dtype=F32, device=DEV)) for p in range(P)])
    rank = 7                                              # the last rank: its shard holds the ragged tail (4088 real rows)
    lo, hi = rank * S, L
    q_r = c["qh"][:, lo:hi].contiguous()
    qm = dit.head_max_norm2(q_r, hi - lo, torch.empty(H, dtype=F32, device=DEV))
    out = torch.empty((hi - lo, H * 128), dtype=BF, device=DEV)
    with _BodyCounter() as n:
        dit.attention(q_r, k_all, vt_all, out, L, 0.0, kmax2=km_all, qmax2=qm)        # nsplit from dit.kv_splits, as the DiT calls it
    assert (n.tracked, n.untracked) == (0, _wgs(hi - lo, 2)), (n.tracked, n.untracked)
    rows = [lo, lo + 1, lo + 255, lo + 256, lo + 2047, L - 65, L - 64, L - 2, L - 1]
    heads = (0, 11, 39)
    want = _oracle_rows(c, rows, heads)
    _compare(out, want, [r - lo for r in rows], heads, 4e-4, 1e-2, "timed_attn.split_kv_shard")   # measured 2.0e-4 / 5.2e-3
    # and the unsplit single-GPU launch on the same rows: equal up to the re-association of the two partial sums
    full = torch.empty((L, H * 128), dtype=BF, device=DEV)
    dit.attention(c["qh"], c["kh"], c["vt"], full, L, 0.0, nsplit=1, kmax2=c["km"], qmax2=c["qm"])
    within("timed_attn.split_vs_unsplit", (full[lo:hi].float() - out.float()).abs().max().item() / full.float().abs().max().item(), 7e-3)   # measured 3.4e-3


# ---------------------------------------------------------------------------------------------------------------------
# one real-width DiT layer on the C2 token grid against oracle.dit on sampled tokens (a-16 at L = 32 760)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,T,Hh,Ww", [("C2", 21, 60, 104), ("C3", 21, 90, 160)])
def test_dit_one_real_width_layer_vs_oracle_on_sampled_tokens(name, T, Hh, Ww):
    """a-16 at the token counts of BASELINE configs[1] (32 760) and configs[2] (720 x 1280: 75 600; VERDICT r3 missing #5 -- until round 4
    the 720p DiT was only compared with itself, sharded == single)."""
    from worldforge_amd import dit
    ocfg = odit.DiTConfig(num_layers=1)
    W = odit.random_weights(ocfg, seed=31)
    Wb = {k: (v.to(BF).float() if v.dim() >= 2 else v) for k, v in W.items()}
    cfg = dit.DiTConfig.wan_i2v_14b()
    cfg.num_layers = 1
    model = dit.WanTransformer3DModel(cfg, DEV).load_state_dict(W)
    g = torch.Generator().manual_seed(32)
    L = T * (Hh // 2) * (Ww // 2)
    assert L == {"C2": 32760, "C3": 75600}[name]
    x = torch.randn(36, T, Hh, Ww, generator=g).to(BF)
    ctx = (torch.randn(200, 4096, generator=g) * 0.1).to(BF)
    clip = torch.randn(257, 1280, generator=g).to(BF)
    with _BodyCounter() as n:
        out = model.forward_tokens(x.to(DEV), 749.0, ctx.to(DEV), clip.to(DEV)).cpu()
    assert n.untracked == _wgs(L) and n.tracked == 0, (n.tracked, n.untracked)   # the layer's self-attention ran the timed body
    assert torch.isfinite(out).all()
    tpf = (Hh // 2) * (Ww // 2)
    rows = sorted(set([0, 1, Ww // 2 - 1, Ww // 2, tpf - 1, tpf, L - tpf - 1, L - tpf, L - 2, L - 1]
                      + torch.randint(0, L, (64,), generator=g).tolist()))
    assert len(rows) >= 64
    with torch.no_grad():
        want = odit.forward_rows(Wb, ocfg, x.float(), torch.tensor(749), ctx.float(), clip.float(), rows)
    got = odit.token_patches(out, ocfg, rows)
    rel = (got - want).norm().item() / want.norm().item()
    err = (got - want).abs().max().item()
    print(f"one real-width DiT layer ({name}), L = {L}, {len(rows)} sampled tokens: rel L2 {rel:.3e}, max abs {err:.3e} (|want| max {want.abs().max().item():.2f})")
    within(f"dit_layer.{name}.rel_l2", rel, 4.4e-3)   # measured 2.30e-3 (C2), 2.24e-3 (C3)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at full size: one released-width LongCat block at 37 440 tokens; the 98 560-token block-sparse attention
# ---------------------------------------------------------------------------------------------------------------------
def test_longcat_released_width_block_37440_tokens_vs_oracle_on_sampled_tokens():
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    ocfg = olc.LongCatConfig(depth=1)
    cfg = LongCatConfig(depth=1)
    assert (cfg.hidden_size, cfg.num_heads, cfg.ffn_hidden, cfg.caption_channels) == (4096, 32, 11008, 4096)
    W = olc.random_weights(ocfg, seed=41)
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(W)
    g = torch.Generator().manual_seed(42)
    T, Hh, Ww = 24, 60, 104                                   # 93 frames of 480 x 832 (bench.py --workload longcat)
    tpf = (Hh // 2) * (Ww // 2)
    L = T * tpf
    assert L == 37440
    x = torch.randn(16, T, Hh, Ww, generator=g).to(BF)
    cap = torch.randn(64, 4096, generator=g).to(BF)
    mask = torch.zeros(64, dtype=torch.int64)
    mask[:50] = 1
    ts = [0.0] + [500.0] * (T - 1)
    got_full = m.forward_tokens(x.to(DEV), ts, cap.to(DEV), mask, 1).float().cpu()
    assert torch.isfinite(got_full).all()
    rows = sorted(set([0, 1, tpf - 1, tpf, tpf + 1, 2 * tpf - 1, L - tpf, L - 2, L - 1]            # condition frame, first / last noise frames
                      + torch.randint(0, tpf, (8,), generator=g).tolist() + torch.randint(tpf, L, (48,), generator=g).tolist()))
    with torch.no_grad():
        want = olc.forward_rows(W, ocfg, x.float(), torch.tensor(ts), cap.float(), mask, 1, rows)
    got = olc.token_patches(got_full, ocfg, rows)
    rel = (got - want).norm().item() / want.norm().item()
    print(f"released-width LongCat block, L = {L}, {len(rows)} sampled tokens: rel L2 {rel:.3e}, max abs {(got - want).abs().max().item():.3e}")
    within("longcat_block.37440.rel_l2", rel, 7.2e-3)   # measured 3.61e-3


def test_block_sparse_attention_98560_tokens_product_selection_vs_oracle_on_sampled_query_blocks():
    """The refine pass's attention at its real size (704 x 1280 x 109 frames -> 98 560 tokens = 770 blocks of 128, 32 heads, sparsity
    0.875 -> 96 key blocks per query block): mean pool -> batched block scores -> wf_bsa_topk_lists -> wf_attn_bsa_fwd, against
    oracle/bsa.py (bsa_interface.py:169-224, 538-560) on sampled (head, query block) pairs: the SELECTION (bf16 gating, as the
    reference's bf16 model) and the attention rows computed with it."""
    from worldforge_amd import bsa
    Hh, S, blk, sparsity = 32, 98560, 128, 0.875
    nb = S // blk
    q = _dev_randn((Hh, S, 128), 1001)
    # keys with a per-block offset so that block scores are well separated (a random field has 770 nearly tied scores per row)
    k = (_dev_randn((Hh, S, 128), 1002).float() + 0.5 * _dev_randn((Hh, nb, 1, 128), 1003).float().expand(Hh, nb, blk, 128).reshape(Hh, S, 128)).to(BF)
    v = _dev_randn((Hh, S, 128), 1004)
    vt = v.view(Hh, S // 64, 64, 128).transpose(2, 3).contiguous()
    out = torch.full((S, Hh * 128), float("nan"), dtype=BF, device=DEV)
    sc = bsa.block_scores(bsa.mean_pool(q, blk), bsa.mean_pool(k, blk))
    sel = bsa.sparse_attention_topk(q, k, vt, out, sc, sparsity, 128 ** -0.5, blk)
    torch.cuda.synchronize()
    idx = sel.cpu()                                                       # [heads, 770, 96] ascending
    n_sel = int((1 - sparsity) * nb)
    assert idx.shape == (Hh, nb, n_sel) and n_sel == 96
    assert not torch.isnan(out.float()).any()
    for h in (0, 13, 31):
        qh, kh, vh = q[h].cpu(), k[h].cpu(), v[h].cpu()
        # gating (bsa_interface.py:169-185) in bf16, as the reference's bf16 model runs it: pooled means and block scores
        osc = torch.matmul(obsa.mean_pool(qh[None], blk), obsa.mean_pool(kh[None], blk).transpose(-1, -2))[0].float()
        psc = sc[h].float().cpu()
        within("bsa.block_scores", (psc - osc).abs().max().item() / osc.abs().max().item(), 1.7e-3)       # measured 8.4e-4 (a fraction of a bf16 ulp at the largest score)
        # selection (bsa_interface.py:211-224): a VALID top-96 of the product's own scores (770 bf16 scores per row tie at the 96th place
        # all the time, and the reference's torch.topk breaks ties arbitrarily) ...
        chosen = torch.zeros((nb, nb), dtype=torch.bool).scatter_(1, idx[h], True)
        assert (chosen.sum(-1) == n_sel).all()
        worst_in = psc.masked_fill(~chosen, float("inf")).min(-1).values
        best_out = psc.masked_fill(chosen, float("-inf")).max(-1).values
        assert (worst_in >= best_out).all()
        # ... and exactly torch.topk's set wherever the boundary is not tied
        strict = worst_in > best_out
        tk = torch.zeros((nb, nb), dtype=torch.bool).scatter_(1, torch.topk(psc, n_sel)[1], True)
        assert (tk[strict] == chosen[strict]).all() and strict.float().mean().item() > 0.2
        for b in (0, 1, 384, 385, nb - 2, nb - 1):
            rows = slice(b * blk, (b + 1) * blk)
            # flash_attn_bsa_varlen_mask.py:236-285 on the selected key blocks (oracle/bsa.py sparse_attention, one query block)
            keys = torch.cat([torch.arange(j * blk, (j + 1) * blk) for j in idx[h, b].tolist()])
            want = obsa.sparse_attention(qh[None, rows].float(), kh[None, keys].float(), vh[None, keys].float(),
                                         torch.arange(n_sel).view(1, 1, n_sel), blk, blk, 128 ** -0.5)[0]
            got = out[rows, h * 128:(h + 1) * 128].float().cpu()
            err = (got - want).abs().max().item()
            within("bsa.rows_vs_oracle", err / (1e-2 * want.abs().max().item() + 2e-3), 0.2)   # measured 0.099 of the bf16-P bar
