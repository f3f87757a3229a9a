"""Seeded inputs shared by tools/record_thirdparty_goldens.py (which records what opencv-python / pytorch3d return for them) and the tests
that switch on when those files exist (tests/test_thirdparty_goldens.py, tests/test_gpu_thirdparty_goldens.py).  numpy's legacy RandomState
streams are frozen, so the inputs regenerate identically on any machine and the golden files carry outputs only."""
import numpy as np

# name: (seed, channels, frames, h, w); "latent" is the FLF gate's real geometry (16 channels of a 60 x 104 latent, scheduling_unipc_multistep_clean.py:373-397)
FARNEBACK_CASES = {"latent": (11, 16, 5, 60, 104), "odd": (12, 3, 3, 45, 70)}


def farneback_latents(name):
    """float32 [C, T, h, w]: smooth moving structure + noise, the kind of field a latent channel is (a pure noise field has no flow)."""
    seed, C, T, h, w = FARNEBACK_CASES[name]
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((C, T, h, w), dtype=np.float32)
    for c in range(C):
        ph = rs.uniform(0, 6.28, size=4)
        vx, vy = rs.uniform(-2.5, 2.5, size=2)
        amp = rs.uniform(0.5, 2.0)
        for t in range(T):
            out[c, t] = amp * (np.sin((xx - vx * t) / 7.0 + ph[0]) * np.cos((yy - vy * t) / 5.0 + ph[1])
                               + 0.5 * np.sin((xx + yy - (vx + vy) * t) / 11.0 + ph[2])) + 0.05 * rs.standard_normal((h, w))
    return out.astype(np.float32)


def farneback_frames(name):
    """uint8 [C, T, h, w]: the reference's preparation of the latent clip (SCHED:373-388 global min / max over ALL channels, :175-176 x255 and
    truncation) -- plain float32 numpy, no third-party arithmetic."""
    x = farneback_latents(name)
    gmin, gmax = np.float32(x.min()), np.float32(x.max())
    norm = ((x - gmin) / np.float32(gmax - gmin)).astype(np.float32)
    return (norm * np.float32(255)).astype(np.float32).astype(np.uint8)


def crackfill_inputs(seed=21, H=96, W=128):
    """image f32 [H, W, 3] in [0, 1], mask u8 [H, W] with thin cracks and a few larger holes (what a forward splat leaves)."""
    rs = np.random.RandomState(seed)
    img = rs.uniform(0, 1, size=(H, W, 3)).astype(np.float32)
    mask = np.ones((H, W), dtype=np.uint8)
    mask[rs.uniform(size=(H, W)) < 0.08] = 0
    for _ in range(6):
        y, x = rs.randint(0, H - 8), rs.randint(0, W - 8)
        mask[y:y + rs.randint(2, 8), x:x + rs.randint(2, 8)] = 0
    mask[:, 0] = rs.randint(0, 2, size=H)      # activity on the border rows / columns: the border modes are what is being pinned
    mask[0, :] = rs.randint(0, 2, size=W)
    return img, mask


def pointrender_inputs(seed=31, H=64, W=96):
    """-> points f32 [H*W, 3] (a depth map un-projected as warp_depthcrafter.py:259-264 does), extrinsic f32 [4, 4] (a small camera move),
    K f32 [3, 3], (H, W), depth f32 [H, W]."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    disp = 0.4 + 0.3 * np.sin(xx / 13.0) * np.cos(yy / 9.0) + 0.3 * (xx > W / 2)      # a depth step: disocclusion when the camera moves
    depth = (1.0 / (disp + 0.1)).astype(np.float32)
    K = np.array([[525.0, 0, W / 2], [0, 525.0, H / 2], [0, 0, 1]], dtype=np.float32)
    x = (xx - K[0, 2]) / K[0, 0] * depth
    y = (yy - K[1, 2]) / K[1, 1] * depth
    pts = np.stack([x, y, depth], axis=-1).reshape(-1, 3).astype(np.float32)
    ang = 0.03
    ext = np.eye(4, dtype=np.float32)
    ext[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=np.float32)
    ext[:3, 3] = np.array([0.02, -0.01, 0.03], dtype=np.float32) + 0.0 * rs.standard_normal(3).astype(np.float32)
    return pts, ext, K, (H, W), depth
