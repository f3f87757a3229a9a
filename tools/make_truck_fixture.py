"""Build tests/golden/truck/: a 9-frame subset of the reference's own test case (test_case/truck/imgs: 49 warped frames + 49 validity
masks of 1280 x 720 from the VGGT warper) for BASELINE config 1 ("8 frames -> 9, 4 steps, IRR only, plumbing").
Every 6th frame (0 .. 20 degrees of the camera path), down-sized to 256 x 144 so the fixture stays < 1 MB.  Data only (PNG pixels);
runs in the build container, where /root/reference exists.

    python tools/make_truck_fixture.py
"""
import os

from PIL import Image

SRC = "/root/reference/test_case/truck/imgs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "truck")
SIZE = (256, 144)

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    warps = sorted(f for f in os.listdir(SRC) if f.startswith("warp_"))
    masks = sorted(f for f in os.listdir(SRC) if f.startswith("mask_"))
    assert len(warps) == len(masks) == 49
    for i in range(0, 49, 6):
        Image.open(os.path.join(SRC, warps[i])).convert("RGB").resize(SIZE, Image.LANCZOS).save(os.path.join(DST, warps[i]), optimize=True)
        Image.open(os.path.join(SRC, masks[i])).convert("L").resize(SIZE, Image.NEAREST).save(os.path.join(DST, masks[i]), optimize=True)
    print(sorted(os.listdir(DST)), sum(os.path.getsize(os.path.join(DST, f)) for f in os.listdir(DST)))
