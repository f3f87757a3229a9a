"""GPU: the whole guided job (IRR + FLF + DSG sampler, DiT, real-config VAE) on the HIP path against the CPU oracle (fp32) run with
identical weights, seeds and inputs -- SURVEY 8d's "PSNR vs reference on final frames" (target >= 40 dB).  Tolerance: the
product computes the DiT's GEMM / attention operands in bf16 (as the reference does on a GPU) and the VAE in its fp32-class mode (the
reference's VAE is fp32); the oracle computes everything in fp32.  Jobs this short never reach an FLF swap (current_step <= 5 -> no
channel): the schedule-length behaviour, where the discrete gate decisions matter, is tests/test_gpu_schedule_length.py."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    dict(dim=256, ffn_dim=512, heads=2, layers=2, Fr=9, H=32, Wd=32, steps=3, guide=2),
    dict(dim=1024, ffn_dim=2048, heads=8, layers=4, Fr=13, H=64, Wd=96, steps=5, guide=3),
    dict(dim=5120, ffn_dim=13824, heads=40, layers=2, Fr=9, H=64, Wd=96, steps=4, guide=3),   # the 14B width, two layers
    dict(dim=1024, ffn_dim=2048, heads=8, layers=4, Fr=13, H=64, Wd=96, steps=5, guide=3, vae_precision="bf16"),  # the opt-in fast VAE
    dict(dim=1024, ffn_dim=2048, heads=8, layers=4, Fr=13, H=64, Wd=96, steps=5, guide=3, vae_precision="fp16"),  # one-term fp16 = TF32-class
])
def test_guided_job_frames_psnr_vs_oracle(cfg):
    import __graft_entry__ as ge
    psnr, err = ge.parity_run(**cfg)
    print(f"PSNR {psnr:.1f} dB, max abs err {err:.4f}")
    assert psnr >= 40.0, (psnr, err)


@pytest.mark.parametrize("cfg", [
    dict(hidden=256, heads=2, depth=2, Fr=9, H=32, Wd=32, steps=4, guide=3),
    dict(hidden=1024, heads=8, depth=3, Fr=13, H=64, Wd=96, steps=5, guide=4, resample=3),
    # the released width, one block.  Temporal-difference FLF: on this 8 x 8 latent grid with random weights the Farneback similarities of
    # the 16 channels lie within bf16 noise of each other, so the (discrete) channel choice of product and oracle differs -- the same
    # chaos DESIGN.md section 4b documents for Wan; the Farneback gate itself is pinned in test_gpu_flow.py
    dict(hidden=4096, heads=32, depth=1, Fr=9, H=64, Wd=64, steps=4, guide=3, flow_backend="tdiff"),
    # the VAE as the LongCat entry loads it (run_longcat_worldforge_single.py:205): a bf16 module, bf16 matrix operands -- the defaults of
    # bench.py's LongCat windows since round 6
    dict(hidden=256, heads=2, depth=2, Fr=9, H=32, Wd=32, steps=4, guide=3, vae_precision="bf16", vae_dtype="bf16"),
    dict(hidden=1024, heads=8, depth=3, Fr=13, H=64, Wd=96, steps=5, guide=4, resample=3, vae_precision="bf16", vae_dtype="bf16"),
])
def test_longcat_guided_job_frames_psnr_vs_oracle(cfg):
    """LongCat-Video config of SURVEY section 8f-1: Euler flow-match sampler + IRR / FLF / DSG / CFG-zero + LongCat DiT + VAE."""
    import __graft_entry__ as ge
    psnr, err = ge.longcat_parity_run(**cfg)
    print(f"LongCat PSNR {psnr:.1f} dB, max abs err {err:.4f}")
    assert psnr >= 40.0, (psnr, err)


@pytest.mark.parametrize("vae", [dict(), dict(vae_precision="bf16", vae_dtype="bf16")])
def test_longcat_refine_pass_frames_psnr_vs_oracle(vae):
    """the 720p refine pass end to end (block-sparse DiT + VAE + up-sampling) against the CPU oracle, which makes
    its own block selection (bf16 gating): blocks at the top-k margin may differ, so the bar is the path's 40 dB, not bit parity."""
    import __graft_entry__ as ge
    psnr, err = ge.longcat_refine_parity_run(hidden=256, heads=2, depth=2, F0=5, H0=48, W0=64, H=128, Wd=128, steps=6, **vae)
    print(f"LongCat refine PSNR {psnr:.1f} dB, max abs err {err:.4f}")
    assert psnr >= 40.0, (psnr, err)


def test_guided_job_with_a_one_sided_hole_is_bit_identical_with_and_without_the_needed_columns_decode():
    """Round 6: the IRR injection decodes only the pixel columns its blend can see (vae.decode(columns=...)).  A whole guided job (IRR + FLF +
    DSG, CFG) on a mask whose hole sits on the right (SURVEY 8d's shape) must give the SAME frames, bit for bit, with the crop on and off --
    and still clear the 40 dB bar against the CPU oracle."""
    import torch

    import __graft_entry__ as ge
    from oracle import dit as odit
    from oracle import vae as ovae
    from worldforge_amd import dit
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    Fr, H, Wd = 9, 64, 256
    g = torch.Generator().manual_seed(7)
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    ref[:, :, 0] = image
    xs = torch.arange(Wd).view(1, 1, 1, 1, Wd).float()
    fr = torch.arange(Fr).view(1, 1, Fr, 1, 1).float() / (Fr - 1)
    mask = (xs < Wd * (1 - 0.35 * fr)).float().expand(1, 1, Fr, H, Wd).contiguous()
    psnr, err = ge.parity_run(dim=256, ffn_dim=512, heads=2, layers=2, Fr=Fr, H=H, Wd=Wd, steps=3, guide=2, inputs=(image, ref, mask))
    print(f"PSNR {psnr:.1f} dB")
    assert psnr >= 40.0
    # the same job twice on the product path, crop on / off
    dev = torch.device("cuda:0")
    ocfg = odit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    Wd_ = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in odit.random_weights(ocfg, seed=3).items()}
    model = dit.WanTransformer3DModel(dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64), dev).load_state_dict(Wd_)
    vae = AutoencoderKLWan(dev).load_state_dict(ovae.random_weights(seed=4))
    text = (torch.randn(1, 24, 64, generator=g) * 0.5).to(torch.bfloat16)
    neg = (torch.randn(1, 24, 64, generator=g) * 0.5).to(torch.bfloat16)
    img = torch.randn(1, 257, 1280, generator=g).to(torch.bfloat16)
    frames = []
    for crop in (True, False):
        vae.crop_to_mask = crop
        pipe = WanImageToVideoPipeline(model, vae, UniPCMultistepScheduler(flow_shift=3.0), device=dev)
        out = pipe(image=image, height=H, width=Wd, num_frames=Fr, num_inference_steps=3, guidance_scale=4.0, generator=torch.manual_seed(42),
                   prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img, output_type="np", video_ref=ref, mask=mask, static=True,
                   guided=True, resample_steps=2, guide_steps=2, omega=4.0, omega_resample=4.0, resample_round=2, use_pca_channel_selection=True)
        frames.append(torch.from_numpy(out.frames))
        if crop:
            cols = vae.needed_columns(next(iter(vae._cols_cache.values()))[0])
            assert vae._crop_range(cols, Wd // 8) is not None          # the crop really happened
    vae.crop_to_mask = True
    assert torch.equal(frames[0], frames[1])
