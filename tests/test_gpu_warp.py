"""GPU: stage-1 forward splat (wf_warp_splat) against the reference's own outputs (golden g16) and the numpy oracle.
Integer / byte work: masks (which target pixels are hit) must be bit-exact and the z-buffer equal to fp32 rounding.  Colours must be
bit-exact wherever the nearest source is unique; where two sources reach the same target with z equal to the last fp64 bits (flat-depth
regions) the winner depends on the evaluation order of the reference's BLAS, so there the test only demands that the colour written is
that of A source pixel with the winning depth: <= 0.5 % of the pixels, each checked through the z-buffer."""
import os

import numpy as np
import pytest
import torch

from oracle import warp as owarp

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g16_warp.npz"))


def _check(imgs, masks, depths, wi, wm, wd):
    imgs, masks, depths = imgs.cpu().numpy(), masks.cpu().numpy(), depths.cpu().numpy()
    assert np.array_equal(masks, wm)
    assert np.array_equal(np.isnan(depths), np.isnan(wd))
    assert np.allclose(np.nan_to_num(depths), np.nan_to_num(wd), rtol=1e-6, atol=0)   # the same surface won everywhere
    diff = (imgs != wi).any(-1)
    assert diff.mean() <= 0.005, diff.mean()
    assert not diff[masks == 0].any()


@pytest.mark.parametrize("name", ["right", "forward"])
def test_splat_equals_reference_golden(name):
    from worldforge_amd import warp
    cams = list(G[f"{name}_cams"])[1:]
    imgs, masks, depths = warp.forward_splat(G["image"], G["depth"], G["K"], G["E"], cams)
    _, _, wd = owarp.splat(G["image"], G["depth"], G["K"], G["E"], cams)
    _check(imgs, masks, depths, G[f"{name}_imgs"][1:], G[f"{name}_masks"][1:], wd)


def test_splat_at_video_size_matches_oracle():
    from worldforge_amd import warp
    rng = np.random.default_rng(2)
    H, W = 480, 832
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (3.0 + np.sin(xx / 40.0) + 0.5 * np.cos(yy / 31.0)).astype(np.float32)
    depth[(xx > 500) & (yy > 200)] = 1.4
    depth[rng.random((H, W)) < 0.02] = np.nan
    image = rng.random((H, W, 3)).astype(np.float32)
    K = np.array([[600.0, 0, W / 2 - 0.5], [0, 600.0, H / 2 - 0.5], [0, 0, 1]])
    cams = []
    for th in (0.02, 0.05, 0.09):
        c = np.eye(4)
        c[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        c[:3, 3] = [-3.0 * np.sin(th), 0.0, 3.0 * (1 - np.cos(th))]
        cams.append(c)
    imgs, masks, depths = warp.forward_splat(image, depth, K, np.eye(4), cams)
    _check(imgs, masks, depths, *owarp.splat(image, depth, K, np.eye(4), cams))


DEV = "cuda:0"


def _case(H=64, W=96):
    """A slanted, textured plane with a nearer box in front of it, a few invalid depths; intrinsics / identity extrinsics."""
    rng = np.random.default_rng(9)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (3.0 + 0.01 * xx + 0.004 * yy).astype(np.float32)
    depth[H // 3:2 * H // 3, W // 3:W // 2] = 1.6
    depth[rng.random((H, W)) < 0.01] = np.nan
    image = rng.random((H, W, 3)).astype(np.float32)
    K = np.array([[80.0, 0, W / 2 - 0.5], [0, 80.0, H / 2 - 0.5], [0, 0, 1]])
    return image, depth, K, np.eye(4)


def _cf_check(img_u8, mask, depth, params):
    """wf_crack_fill vs oracle/crackfill.py on a stack of views: masks bit-exact, depths equal, colours within one grey level (the
    8-neighbour sums are fp32 in a different order)."""
    from oracle import crackfill as ocf
    from worldforge_amd import warp
    oi, om, od = warp.crack_fill(torch.from_numpy(img_u8).to(DEV), torch.from_numpy(mask).to(DEV), torch.from_numpy(depth).to(DEV),
                                 min_neighbors=params["min_neighbors"], min_valid_neighbors=params["min_valid_neighbors"])
    oi, om, od = oi.cpu().numpy(), om.cpu().numpy(), od.cpu().numpy()
    filled = 0
    for f in range(img_u8.shape[0]):
        wi, wm, wd = ocf.warp_frame_fill(img_u8[f], mask[f], depth[f], params)
        np.testing.assert_array_equal(om[f], wm)
        assert np.array_equal(np.isnan(od[f]), np.isnan(wd))
        np.testing.assert_allclose(od[f][~np.isnan(wd)], wd[~np.isnan(wd)], rtol=1e-6)
        assert np.abs(oi[f].astype(np.int32) - wi.astype(np.int32)).max() <= 1
        filled += int(((wm > 0) & (mask[f] == 0)).sum())
    return filled


def test_crack_fill_matches_oracle_on_constructed_scenes():
    from oracle import crackfill as ocf
    from tests.test_oracle_crackfill import _scene
    img, mask, depth = _scene()
    stack_i, stack_m, stack_d = [(img * 255).astype(np.uint8)], [mask], [depth]
    # a second view: random holes + stray pixels over three depth layers, so that several segments have outliers and fills overlap
    rng = np.random.default_rng(5)
    H, W = img.shape[:2]
    d2 = np.choose(rng.integers(0, 3, (H // 4, W // 4)).repeat(4, 0).repeat(4, 1), [1.0, 2.5, 6.0]).astype(np.float32) + rng.random((H, W)).astype(np.float32) * 0.2
    m2 = (rng.random((H, W)) > 0.12).astype(np.uint8)
    i2 = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    d2[m2 == 0] = np.nan
    i2[m2 == 0] = 0
    stack_i.append(i2), stack_m.append(m2), stack_d.append(d2)
    # a third view: a single depth value (min == max: one segment), and a fourth: empty
    d3 = np.full((H, W), 2.0, dtype=np.float32)
    m3 = m2.copy()
    d3[m3 == 0] = np.nan
    stack_i.append(i2), stack_m.append(m3), stack_d.append(d3)
    for p in (ocf.RUN_WARP_PARAMS, ocf.DEFAULT_PARAMS):
        filled = _cf_check(np.stack(stack_i), np.stack(stack_m), np.stack(stack_d), p)
        assert filled > 20


def test_crack_fill_after_the_splat_of_a_camera_path():
    """The stage-1 chain on the GPU: forward splat of 4 views of a slanted textured plane, then crack filling, against the oracle chain."""
    from oracle import crackfill as ocf
    from oracle import warp as owarp
    from worldforge_amd import warp
    image, depth, K, E = _case(H=64, W=96)
    cams = warp.camera_path("right", E, 12.0, 5, float(np.nanmean(depth)))[1:]
    gi, gm, gd = warp.forward_splat(torch.from_numpy(image), torch.from_numpy(depth), K, E, cams, device=DEV)
    wi, wm, wd = owarp.splat(image, depth, K, E, cams)
    np.testing.assert_array_equal(gm.cpu().numpy(), wm)
    # crack filling is compared on the ORACLE's splat (colour ties of the splat aside), the full GPU chain must agree on the masks
    filled = _cf_check(wi, wm, wd, ocf.RUN_WARP_PARAMS)
    assert filled > 0
    fi, fm, fd = warp.crack_fill(gi, gm, gd, min_valid_neighbors=2)
    want_m = np.stack([ocf.warp_frame_fill(wi[f], wm[f], wd[f], ocf.RUN_WARP_PARAMS)[1] for f in range(len(cams))])
    np.testing.assert_array_equal(fm.cpu().numpy(), want_m)


# ---- DepthCrafter point-cloud renderer (wf_points_render / wf_depth_edge_mask) against oracle/pointrender.py (both unpinned: pytorch3d / cv2) ----
def _pr_scene(H, W, seed=0):
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    disp = (0.3 + 0.2 * np.sin(xx / 40.0) + 0.25 * (yy > H // 2) + 0.15 * ((xx // 64) % 2) + 0.02 * rng.random((H, W))).astype(np.float32)
    depth = (1.0 / (disp + 0.1)).astype(np.float32)
    rgb = rng.random((H, W, 3)).astype(np.float32)
    K = np.array([[525, 0, W / 2], [0, 525, H / 2], [0, 0, 1]], dtype=np.float32)
    return rgb, depth, K


def test_depth_edge_mask_equals_oracle():
    import numpy as np
    from oracle import pointrender as opr
    from worldforge_amd import warp
    for H, W in ((320, 512), (97, 131), (576, 1024)):
        _, depth, _ = _pr_scene(H, W, seed=H)
        got = warp.depth_edge_mask(torch.from_numpy(depth).to(DEV)).cpu().numpy().astype(bool)
        want = opr.edge_filter_mask(depth)
        assert (got != want).mean() <= 1e-5, (H, W, (got != want).sum())   # (a magnitude within an ulp of the threshold may fall either way)
    flat = torch.full((32, 48), 2.0, device=DEV)
    assert not warp.depth_edge_mask(flat).any()


@pytest.mark.parametrize("H,W,move,edge", [(320, 512, 0.0, False), (320, 512, 0.08, True), (576, 1024, -0.05, True), (400, 400, 0.03, False),
                                           (512, 320, 0.04, True)])
def test_points_render_equals_oracle(H, W, move, edge):
    """Same float32 projection sequence on both sides: the nearest-point index per pixel -- hence image and mask -- must agree except where a
    point sits within float rounding of the disc edge (bounded at 1e-4 of the pixels)."""
    import numpy as np
    from oracle import pointrender as opr
    from worldforge_amd import warp
    rgb, depth, K = _pr_scene(H, W, seed=3)
    cam = np.eye(4)
    cam[0, 3] = move
    cam[1, 3] = -move / 2
    th = 0.03 if move else 0.0
    cam[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    img, mask = warp.render_depthcrafter_frame(torch.from_numpy(rgb).to(DEV), torch.from_numpy(depth).to(DEV), cam, K, edge_filter=edge)
    wimg, wmask = opr.render_frame(rgb, depth, cam, K, edge)
    img, mask = img.cpu().numpy(), mask.cpu().numpy()
    assert mask.shape == wmask.shape == (H, W, 1) and img.shape == (H, W, 3)
    assert (mask != wmask).mean() <= 1e-4
    same = (mask == wmask)[..., 0]
    assert (np.abs(img - wimg).max(-1)[same] > 0).mean() <= 1e-4
    assert not img[mask[..., 0] == 0].any()
    if move:
        assert 0.3 < mask.mean() < 1.0


def test_points_render_without_opening_and_with_dropped_points():
    import numpy as np
    from oracle import pointrender as opr
    from worldforge_amd import warp
    H, W = 320, 480
    rgb, depth, K = _pr_scene(H, W, seed=9)
    pts = opr.unproject(depth, K)
    drop = (np.random.default_rng(1).random(H * W) < 0.3)
    cam = np.eye(4)
    cam[0, 3] = 0.02
    img, mask = warp.points_render(torch.from_numpy(pts).to(DEV), torch.from_numpy(rgb.reshape(-1, 3)).to(DEV), cam, K, (H, W), morph=False,
                                   drop=torch.from_numpy(drop.astype(np.uint8)).to(DEV))
    wimg, wmask = opr.project_points_to_image(pts[~drop], rgb.reshape(-1, 3)[~drop], cam, K, (H, W), morph=False)
    assert (mask.cpu().numpy() != wmask).mean() <= 1e-4
    assert (np.abs(img.cpu().numpy() - wimg).max(-1) > 0).mean() <= 2e-4


def test_small_crack_fill_equals_reference_recorded_outputs():
    """wf_fill_small_cracks (the <= 100-splatted-depths path of warp_single_img, utils_warp.py:973-981) against G23 = the reference's own
    fill_small_cracks on that view, 8 parameter sets incl. ones where the depth-guided sequential step fills pixels: masks exact, colours
    within one grey level after the single (x * 255) quantisation."""
    from oracle import crackfill as ocf
    from tests.cases import SMALL_CRACK_CASES
    from worldforge_amd import warp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g23_fill_small_cracks.npz"))
    img_u8 = (g["img"] * 255).astype(np.uint8)
    img_f = img_u8.astype(np.float32) / 255.0            # the reference's float image of the u8 view (:956)
    mask, depth = g["mask"], g["depth"]
    for name, (has_conf, mcs, mvn, thr) in SMALL_CRACK_CASES.items():
        want_i, want_m = ocf.fill_small_cracks_depth_guided(img_f, mask, depth, has_conf, thr, mcs, mvn)
        assert np.array_equal(want_m, g[f"{name}_mask"]), name       # same fill decisions as the recorded run (colours differ: u8 input here)
        oi, om = warp.small_crack_fill(torch.from_numpy(img_u8).to(DEV), torch.from_numpy(mask).to(DEV), torch.from_numpy(depth).to(DEV),
                                       has_conf, thr, mcs, mvn)
        assert np.array_equal(om.cpu().numpy(), g[f"{name}_mask"]), name
        want_u8 = (want_i * 255).astype(np.uint8)
        assert np.abs(oi.cpu().numpy().astype(np.int32) - want_u8.astype(np.int32)).max() <= 1, name


def test_crack_fill_routes_nearly_empty_views_through_the_small_crack_path():
    """A camera path whose last view has <= 100 splatted pixels: crack_fill must return fill_small_cracks' result for it (depth untouched)
    and the depth-aware result for the others."""
    from oracle import crackfill as ocf
    from worldforge_amd import warp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g23_fill_small_cracks.npz"))
    H, W = g["mask"].shape
    rs = np.random.default_rng(3)
    full_img = (rs.random((H, W, 3)) * 255).astype(np.uint8)
    full_mask = (rs.random((H, W)) > 0.1).astype(np.uint8)
    full_depth = np.where(full_mask > 0, 2.0 + rs.random((H, W)), np.nan).astype(np.float32)
    sparse_img = (g["img"] * 255).astype(np.uint8)
    sparse_mask = g["mask"]
    sparse_depth = np.where(sparse_mask > 0, g["depth"], np.nan).astype(np.float32)
    imgs = torch.from_numpy(np.stack([full_img, sparse_img])).to(DEV)
    masks = torch.from_numpy(np.stack([full_mask, sparse_mask])).to(DEV)
    depths = torch.from_numpy(np.stack([full_depth, sparse_depth])).to(DEV)
    od = torch.from_numpy(g["depth"]).to(DEV)
    oi, om, odp = warp.crack_fill(imgs, masks, depths, min_valid_neighbors=7, original_depth=od, has_depth_conf=True)
    wi, wm, wd = ocf.warp_frame_fill(sparse_img, sparse_mask, sparse_depth, dict(min_valid_neighbors=7), original_depth=g["depth"], use_depth_conf=True)
    assert np.array_equal(om[1].cpu().numpy(), wm) and int(wm.sum()) > int(sparse_mask.sum())
    assert np.abs(oi[1].cpu().numpy().astype(np.int32) - wi.astype(np.int32)).max() <= 1
    assert torch.equal(torch.isnan(odp[1]), torch.isnan(depths[1]))
    fi, fm, fd = ocf.warp_frame_fill(full_img, full_mask, full_depth, dict(min_valid_neighbors=7))
    assert np.array_equal(om[0].cpu().numpy(), fm)
