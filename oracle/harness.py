"""Oracle: mask softening and size rule of infer_worldforge.py (INFER:105-150, 218-221) in numpy.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
from scipy.ndimage import distance_transform_edt


def soften_mask(mask, d=15, decay="sine"):
    """INFER:105-150, vectorised per frame."""
    out = mask.astype(np.float32).copy()
    ramps = {"linear": lambda t: t, "exponential": lambda t: 1.0 - np.exp(-3.0 * t), "sine": lambda t: np.sin(np.pi / 2 * t),
             "cosine": lambda t: 1.0 - np.cos(np.pi / 2 * t)}
    for i, m in enumerate(mask):
        on = m.astype(bool)
        if on.all() or (~on).all():
            continue
        dist = distance_transform_edt(on)
        sel = on & (dist <= d)
        fr = m.astype(np.float32).copy()
        fr[sel] = ramps[decay](np.clip(dist[sel] / d, 0.0, 1.0))
        out[i] = fr
    return out


def size_rule(ih, iw, max_area, mod=16):
    """INFER:218-221."""
    ar = ih / iw
    return int(round(np.sqrt(max_area * ar)) // mod * mod), int(round(np.sqrt(max_area / ar)) // mod * mod)
