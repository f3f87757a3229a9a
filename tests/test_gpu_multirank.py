"""GPU: the N > 1 code paths (token-sharded DiT with per-layer K / V^T all-gather and segment-addressed attention, row-slab
sharded VAE with halo all-gather, and the whole guided sampler on top of both) run as P simulated ranks -- threads of one
process on the single test GPU, exchanging through tests/fakes.SimComm -- and compared with the single-rank result.

The real communicator (parallel.Comm over RCCL) is covered by the world-1 RCCL test in test_gpu_dit.py and the world-2 gloo
tests in test_parallel_cpu.py; what this file pins is that the *sharded arithmetic* equals the unsharded one."""
import threading

import pytest
import torch

from tests.fakes import SimComm

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _run_ranks(P, fn):
    shared = {"slots": [None] * P, "bar": threading.Barrier(P)}
    res, errs = [None] * P, []

    def worker(r):
        try:
            res[r] = fn(SimComm(P, r, shared))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    return res


EXCHANGES = [("gather", 1), ("chunked", 1), ("chunked", 2), ("chunked", 3), ("bcast", 1)]
EXCHANGE_IDS = ["gather", "own_first_1_gather", "chunked2", "chunked3", "bcast"]


@pytest.mark.parametrize("exchange", EXCHANGES, ids=EXCHANGE_IDS)
@pytest.mark.parametrize("P,thw", [(2, (3, 16, 20)), (3, (5, 16, 24)), (4, (5, 16, 24)), (8, (5, 16, 24))])
def test_token_sharded_dit_equals_single(P, thw, exchange):
    """Every rank must return the full velocity tensor of the single-rank forward.  The K tiles of the gathered exchange buffer
    ([P] packed slots [K | V^T | bounds], parallel.KVExchange) coincide with the single-rank tiles (all shards but the last are full
    multiples of 64), so the online softmax sees the same sequence of tiles: bit-identical in mode "gather" (one all-gather, one
    launch).  The OWN-FIRST modes (round 5: "chunked" = G all-gathers of the g-th 1/G of every rank's keys, "bcast" = per-source
    broadcasts; the attention walks its own shard first and the peers' windows as they arrive, partial results merged) re-associate
    the fp32 partial sums: equal to the split-KV class of tolerance, and every rank still returns the same tensor (each row comes
    from its owner)."""
    from tests._tol import within
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    T, Hh, Ww = thw
    x = _rand((36, T, Hh, Ww), 60).to(BF).to(DEV)
    ctx, clip = _rand((30, 64), 61).to(BF).to(DEV), _rand((257, 1280), 62).to(BF).to(DEV)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref = m0.forward_tokens(x, 500.0, ctx, clip).clone()

    def rank_fn(comm):
        m = dit.WanTransformer3DModel(cfg, DEV, comm=comm)
        m.w = m0.w
        m.exchange_mode, m.exchange_chunks = exchange
        assert m.local_tokens(T * (Hh // 2) * (Ww // 2)) > 0
        return m.forward_tokens(x, 500.0, ctx, clip).clone()

    res = _run_ranks(P, rank_fn)
    for r, got in enumerate(res):
        if exchange[0] == "gather":
            assert torch.equal(got, ref), (r, (got - ref).abs().max())
        else:
            assert torch.equal(got, res[0]), r
            within(f"multirank.dit_{exchange[0]}{exchange[1]}.P{P}", (got - ref).abs().max().item() / ref.abs().max().item(), 5.4e-3)   # measured 2.1e-3 ... 2.7e-3


@pytest.mark.parametrize("P", [2, 4])
def test_guided_sampler_on_sharded_dit_and_vae_equals_single(P):
    """End to end: IRR + FLF + DSG sampler, every rank drawing the same CPU-generator noise (no broadcast inside the loop),
    DiT token-sharded, VAE row-sharded.  All ranks must produce the single-rank frames."""
    from worldforge_amd import dit
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    Fr, H, Wd = 9, 128, 160
    g = torch.Generator().manual_seed(3)
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    mask = (torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float()
    text, neg = _rand((1, 30, 64), 7).to(BF), _rand((1, 30, 64), 8).to(BF)
    img = _rand((1, 257, 1280), 9).to(BF)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    v0 = AutoencoderKLWan(DEV).init_random(seed=1)
    kw = dict(guided=True, resample_steps=2, guide_steps=3, omega=4.0, omega_resample=4.0, resample_round=3,
              use_pca_channel_selection=True)

    def run(comm):
        m = dit.WanTransformer3DModel(cfg, DEV, comm=comm)
        m.w = m0.w
        v = AutoencoderKLWan(DEV, comm=comm)
        v.w = v0.w
        if comm is not None:
            assert v.can_shard(H // 8)
        pipe = WanImageToVideoPipeline(m, v, UniPCMultistepScheduler(flow_shift=3.0), device=DEV)
        out = pipe(image=image, height=H, width=Wd, num_frames=Fr, num_inference_steps=4, guidance_scale=4.0,
                   generator=torch.Generator().manual_seed(42), prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img,
                   output_type="np", video_ref=ref, mask=mask, static=True, **kw)
        return torch.from_numpy(out.frames).clone()

    want = run(None)
    assert torch.isfinite(want).all()
    for r, got in enumerate(_run_ranks(P, run)):
        assert torch.equal(got, want), (r, (got - want).abs().max())


@pytest.mark.parametrize("P", [4, 8])
def test_cfg_groups_by_sequence_shards_equals_single(P):
    """SURVEY 8e "P = 8 = 2 x 4": the job's ranks split into two CFG groups (parallel.Comm.split(2)); group 0 runs the positive-prompt
    forward, group 1 the negative one, each sequence-parallel over its P / 2 ranks; one all-gather over the whole job hands every rank
    both velocities (pipeline.cfg_split); the VAE stays row-sharded over all P ranks.  With the one-event exchange inside a group every
    rank must produce the single-GPU frames bit for bit."""
    from worldforge_amd import dit
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    Fr, H, Wd = 9, 128, 160
    g = torch.Generator().manual_seed(3)
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    mask = (torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float()
    text, neg = _rand((1, 30, 64), 7).to(BF), _rand((1, 30, 64), 8).to(BF)
    img = _rand((1, 257, 1280), 9).to(BF)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    v0 = AutoencoderKLWan(DEV).init_random(seed=1)
    kw = dict(guided=True, resample_steps=2, guide_steps=3, omega=4.0, omega_resample=4.0, resample_round=3,
              use_pca_channel_selection=True)

    def run(world):
        sub = world.split(2) if world is not None else None
        m = dit.WanTransformer3DModel(cfg, DEV, comm=sub)
        m.w = m0.w
        m.exchange_mode = "gather"
        v = AutoencoderKLWan(DEV, comm=world)
        v.w = v0.w
        pipe = WanImageToVideoPipeline(m, v, UniPCMultistepScheduler(flow_shift=3.0), device=DEV)
        if world is not None:
            assert (sub.world, sub.rank, sub.group_index) == (P // 2, world.rank % (P // 2), world.rank // (P // 2))
            pipe.cfg_split = (world, sub.group_index)
        out = pipe(image=image, height=H, width=Wd, num_frames=Fr, num_inference_steps=4, guidance_scale=4.0,
                   generator=torch.Generator().manual_seed(42), prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img,
                   output_type="np", video_ref=ref, mask=mask, static=True, **kw)
        return torch.from_numpy(out.frames).clone()

    want = run(None)
    assert torch.isfinite(want).all()
    for r, got in enumerate(_run_ranks(P, run)):
        assert torch.equal(got, want), (r, (got - want).abs().max())


@pytest.mark.parametrize("P,precision", [(2, "bf16x3"), (4, "bf16"), (8, "bf16x3")])
def test_sharded_decode_blend_encode_never_gathers_pixels_and_equals_single(P, precision):
    """vae.decode_blend_encode (the IRR pixel round trip, SCHED:1285-1384) on P row slabs: every rank's posterior mean must equal the
    single-GPU decode -> ops.blend_pixels -> encode bit for bit, and no all-gather may carry a full-resolution frame."""
    from worldforge_amd import ops
    from worldforge_amd.vae import AutoencoderKLWan
    T, h, w = 3, 16, 20
    Fr, H, Wd = 4 * (T - 1) + 1, 8 * h, 8 * w
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, 16, T, h, w, generator=g).to(DEV)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g).to(DEV)
    mask = ((torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float() * torch.rand(1, 1, Fr, H, Wd, generator=g)).to(DEV)
    v0 = AutoencoderKLWan(DEV, precision=precision).init_random(seed=1)
    dec = v0.decode(z, return_dict=False)[0]
    want = v0.encode(ops.blend_pixels(ref, mask, dec)).latent_dist.mode()
    assert torch.equal(v0.decode_blend_encode(z, ref, mask).mode(), want)      # one GPU: the same three calls
    pixel_slabs = []

    def rank_fn(comm):
        inner = comm.all_gather

        def counting(out, inp):
            if inp.dim() == 4 and inp.shape[1] * P == H and inp.shape[2] == Wd:   # [T, H/P, W, C]: a full-resolution row slab
                pixel_slabs.append(tuple(inp.shape))
            return inner(out, inp)

        comm.all_gather = counting
        v = AutoencoderKLWan(DEV, comm=comm, precision=precision)
        v.w = v0.w
        assert v.can_shard(h)
        return v.decode_blend_encode(z, ref, mask).mode().clone()

    for r, got in enumerate(_run_ranks(P, rank_fn)):
        assert torch.equal(got, want), (P, r, (got - want).abs().max())
    # what the ranks exchange are halo row pairs and latent-resolution feature maps, never a full-resolution slab (the gathered form
    # all-gathered the decoded [F, H/P, W, 32] slab before the blend)
    assert not pixel_slabs, pixel_slabs


@pytest.mark.parametrize("P", [1, 2, 4])
def test_cfg_pair_lockstep_equals_two_sequential_forwards(P):
    """forward_tokens_pair advances the positive- and negative-prompt forwards one layer apart (so each K / V exchange hides under
    the other branch's layer); each branch must still equal its own stand-alone forward bit for bit."""
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64)
    T, Hh, Ww = 5, 16, 24
    x = _rand((36, T, Hh, Ww), 70).to(BF).to(DEV)
    ca, cb = _rand((30, 64), 71).to(BF).to(DEV), _rand((12, 64), 72).to(BF).to(DEV)
    clip = _rand((257, 1280), 73).to(BF).to(DEV)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref_a = m0.forward_tokens(x, 321.0, ca, clip).clone()
    ref_b = m0.forward_tokens(x, 321.0, cb, clip).clone()
    assert not torch.equal(ref_a, ref_b)

    def rank_fn(comm):
        m = dit.WanTransformer3DModel(cfg, DEV, comm=comm)
        m.w = m0.w
        a, b = m.forward_tokens_pair(x, 321.0, ca, cb, clip, interleave=True)
        return a.clone(), b.clone()

    results = [rank_fn(None)] if P == 1 else _run_ranks(P, rank_fn)
    for r, (a, b) in enumerate(results):
        assert torch.equal(a, ref_a), (r, (a - ref_a).abs().max())
        assert torch.equal(b, ref_b), (r, (b - ref_b).abs().max())


@pytest.mark.parametrize("exchange", EXCHANGES, ids=EXCHANGE_IDS)
@pytest.mark.parametrize("P,thw,ncl", [(2, (3, 16, 20), 1), (3, (5, 16, 24), 1), (4, (5, 16, 24), 0), (8, (8, 16, 32), 2)])
def test_token_sharded_longcat_dit_equals_single(P, thw, ncl, exchange):
    """LongCat DiT, sequence parallel: per-frame AdaLN selected by the global token index (row0), condition / noise split by global
    index (a shard may hold both kinds of rows, or only one), K / V^T shards exchanged through the packed buffers and consumed in
    place.  Mode "gather": every rank must return the single-rank velocity, bit for bit; the own-first modes: the same tensor on every
    rank, within the re-association tolerance (the condition rows' key prefix is walked through the same chunk windows)."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    kw = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    T, Hh, Ww = thw
    x = _rand((16, T, Hh, Ww), 60).to(BF).to(DEV)
    cap = _rand((30, 64), 61).to(BF).to(DEV)
    mask = torch.zeros(30, dtype=torch.int64)
    mask[:21] = 1
    ts = [0.0] * ncl + [500.0] * (T - ncl)
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV).load_state_dict(W)
    ref = m0.forward_tokens(x, ts, cap, mask, ncl).clone()

    def rank_fn(comm):
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, comm=comm)
        m.w = m0.w
        m.exchange_mode, m.exchange_chunks = exchange
        return m.forward_tokens(x, ts, cap, mask, ncl).clone()

    from tests._tol import within
    res = _run_ranks(P, rank_fn)
    for r, got in enumerate(res):
        if exchange[0] == "gather":   # one all-gather, one launch: the gathered tiles are the single-rank tiles, bit for bit
            assert torch.equal(got, ref), (r, (got - ref).abs().max())
        else:               # windows walked own-first and merged: fp32 re-association of the partial sums (bf16 residual stream)
            assert torch.equal(got, res[0]), r
            within(f"multirank.longcat_{exchange[0]}{exchange[1]}.P{P}", (got - ref).abs().max().item() / ref.abs().max().item(), 8e-3)   # measured 3.1e-3 ... 4.0e-3


@pytest.mark.parametrize("P", [2, 4])
def test_longcat_cfg_batch_lockstep_equals_sequential_samples(P):
    """LongCat's CFG batch (pipeline_longcat_video.py:857-866: [negative, positive] on the batch axis) under sequence parallelism: the two
    samples are advanced in lock-step, one block apart (forward_tokens_pair, round 5), each on its own buffers with the one-event
    all-gather -- each sample must equal its own single-rank forward bit for bit, on every rank, also when called through __call__."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    kw = dict(hidden_size=256, depth=3, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    T, Hh, Ww, ncl = 5, 16, 24, 1
    xs = _rand((2, 16, T, Hh, Ww), 60).to(BF).to(DEV)
    caps = _rand((2, 1, 30, 64), 61).to(BF).to(DEV)
    masks = torch.zeros(2, 30, dtype=torch.int64)
    masks[0, :21] = 1
    masks[1, :9] = 1
    tstep = torch.tensor([[0.0] * ncl + [500.0] * (T - ncl)] * 2)
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV).load_state_dict(W)
    ref = m0(xs, tstep, caps, masks, num_cond_latents=ncl).clone()
    assert not torch.equal(ref[0], ref[1])

    def rank_fn(comm):
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, comm=comm)
        m.w = m0.w
        assert m.pair_lockstep
        return m(xs, tstep, caps, masks, num_cond_latents=ncl).clone()

    for r, got in enumerate(_run_ranks(P, rank_fn)):
        assert torch.equal(got, ref), (r, (got - ref).abs().max())


@pytest.mark.parametrize("P", [2, 4])
def test_longcat_guided_sampler_on_sharded_dit_and_vae_equals_single(P):
    """The whole LongCat guided i2v job (Euler sampler, CFG-zero, IRR re-noise, FLF, DSG) with the DiT token-sharded and the VAE
    row-sharded: every rank draws the same CPU-generator noise and must produce the single-rank frames."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    kw = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    Fr, H, Wd = 9, 128, 160
    g = torch.Generator().manual_seed(3)
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    mask = (torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float()
    pe, ne = _rand((1, 1, 20, 64), 7).to(BF), _rand((1, 1, 20, 64), 8).to(BF)
    pm, nm = torch.ones(1, 20, dtype=torch.int64), torch.ones(1, 20, dtype=torch.int64)
    nm[:, 9:] = 0
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV).load_state_dict(W)
    v0 = AutoencoderKLWan(DEV).init_random(seed=1)

    def run(comm):
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, comm=comm)
        m.w = m0.w
        m.exchange_mode = "gather"   # the bit-identical form of the exchange (the own-first ones are compared with a tolerance above);
        # the CFG batches of this job run as lock-step pairs (forward_tokens_pair), which always use it
        v = AutoencoderKLWan(DEV, comm=comm)
        v.w = v0.w
        pipe = LongCatVideoPipeline(v, FlowMatchEulerDiscreteScheduler(shift=3.0), m, device=DEV)
        out = pipe.generate_i2v(image=image, height=H, width=Wd, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                                negative_prompt_attention_mask=nm, num_frames=Fr, num_inference_steps=4, guidance_scale=4.0,
                                generator=torch.Generator().manual_seed(42), video_ref=ref, mask=mask, guided=True, resample_steps=2,
                                guide_steps=3, resample_round=3, use_pca_channel_selection=True, static=True)
        return torch.from_numpy(out).clone()

    want = run(None)
    assert torch.isfinite(want).all()
    for r, got in enumerate(_run_ranks(P, run)):
        assert torch.equal(got, want), (r, (got - want).abs().max())


@pytest.mark.parametrize("P", [2, 4])
def test_longcat_cfg_groups_by_sequence_shards_equals_single(P):
    """The LongCat guided i2v job with CFG (batch [negative, positive]) as two CFG groups x P / 2 sequence shards (longcat_pipeline.cfg_split;
    P = 2: one sample per GPU, no K / V^T exchange at all): every rank must produce the single-GPU frames bit for bit."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    kw = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    Fr, H, Wd = 9, 128, 160
    g = torch.Generator().manual_seed(3)
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    mask = (torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float()
    pe, ne = _rand((1, 1, 20, 64), 7).to(BF), _rand((1, 1, 20, 64), 8).to(BF)
    pm, nm = torch.ones(1, 20, dtype=torch.int64), torch.ones(1, 20, dtype=torch.int64)
    nm[:, 9:] = 0
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV).load_state_dict(W)
    v0 = AutoencoderKLWan(DEV).init_random(seed=1)

    def run(world):
        sub = world.split(2) if world is not None else None
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, comm=sub if (sub is not None and sub.world > 1) else None)
        m.w = m0.w
        m.exchange_mode = "gather"
        v = AutoencoderKLWan(DEV, comm=world)
        v.w = v0.w
        pipe = LongCatVideoPipeline(v, FlowMatchEulerDiscreteScheduler(shift=3.0), m, device=DEV)
        if world is not None:
            pipe.cfg_split = (world, sub.group_index)
        out = pipe.generate_i2v(image=image, height=H, width=Wd, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                                negative_prompt_attention_mask=nm, num_frames=Fr, num_inference_steps=4, guidance_scale=4.0,
                                generator=torch.Generator().manual_seed(42), video_ref=ref, mask=mask, guided=True, resample_steps=2,
                                guide_steps=3, resample_round=3, use_pca_channel_selection=True, static=True)
        return torch.from_numpy(out).clone()

    want = run(None)
    assert torch.isfinite(want).all()
    for r, got in enumerate(_run_ranks(P, run)):
        assert torch.equal(got, want), (r, (got - want).abs().max())


@pytest.mark.parametrize("P,ncl,T", [(2, 4, 8), (4, 4, 8), (3, 0, 12)])
def test_token_sharded_longcat_dit_with_block_sparse_attention_equals_single(P, ncl, T):
    """The refine-pass DiT (block-sparse self-attention, network resident in 3D-block token order), sequence parallel over whole
    256-row query groups: pooled key blocks and K / V^T shards are all-gathered, every rank selects for its own query blocks."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    kw = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    bsa_params = dict(sparsity=0.5, chunk_3d_shape_q=[4, 4, 8], chunk_3d_shape_k=[4, 4, 8])
    Hh, Ww = 16, 32   # 128 tokens per frame: 8 / 12 blocks of 128
    x = _rand((16, T, Hh, Ww), 60).to(BF).to(DEV)
    cap = _rand((30, 64), 61).to(BF).to(DEV)
    ts = [0.0] * ncl + [500.0] * (T - ncl)
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, enable_bsa=True, bsa_params=bsa_params).load_state_dict(W)
    ref = m0.forward_tokens(x, ts, cap, None, ncl).clone()

    def rank_fn(comm):
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, enable_bsa=True, bsa_params=bsa_params, comm=comm)
        m.w = m0.w
        return m.forward_tokens(x, ts, cap, None, ncl).clone()

    for r, got in enumerate(_run_ranks(P, rank_fn)):
        assert torch.equal(got, ref), (r, (got - ref).abs().max())


def test_longcat_refine_pass_on_two_ranks_equals_single():
    """generate_refine end to end with the block-sparse DiT sequence-parallel (shards of whole query groups in block order) and the VAE
    row-sharded: both ranks must reproduce the single-rank frames."""
    from oracle import longcat_dit as olc
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    kw = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(olc.LongCatConfig(**kw), seed=5)
    bsa_params = dict(sparsity=0.5, chunk_3d_shape_q=[4, 4, 8], chunk_3d_shape_k=[4, 4, 8])
    g = torch.Generator().manual_seed(3)
    frames = (torch.rand(5, 48, 64, 3, generator=g) * 255).to(torch.uint8)
    image = torch.rand(3, 128, 128, generator=g)
    pe = _rand((1, 1, 20, 64), 7).to(BF)
    pm = torch.ones(1, 20, dtype=torch.int64)
    m0 = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, enable_bsa=True, bsa_params=bsa_params).load_state_dict(W)
    v0 = AutoencoderKLWan(DEV).init_random(seed=1)

    def run(comm):
        m = LongCatVideoTransformer3DModel(LongCatConfig(**kw), DEV, enable_bsa=True, bsa_params=bsa_params, comm=comm)
        m.w = m0.w
        v = AutoencoderKLWan(DEV, comm=comm)
        v.w = v0.w
        pipe = LongCatVideoPipeline(v, FlowMatchEulerDiscreteScheduler(shift=3.0), m, device=DEV)
        out = pipe.generate_refine(stage1_video=frames, height=128, width=128, prompt_embeds=pe, prompt_attention_mask=pm, image=image,
                                   num_cond_frames=1, num_inference_steps=6, generator=torch.Generator().manual_seed(42), t_thresh=0.5,
                                   spatial_refine_only=True)
        return torch.from_numpy(out).clone()

    want = run(None)
    assert torch.isfinite(want).all()
    for r, got in enumerate(_run_ranks(2, run)):
        assert torch.equal(got, want), (r, (got - want).abs().max())
