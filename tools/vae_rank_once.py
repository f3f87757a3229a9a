"""One rank of a P-rank job (parallel.LoopbackComm) running the row-sharded VAE decode and encode at C2 a few times: the process
`rocprofv3 --kernel-trace --stats` is pointed at to see where the sharded VAE's time goes (tools/gpurun_scripts/r6_c.sh).
  P=8 RANK_SIM=4 PREC=fp16x3 CROP=1 python tools/vae_rank_once.py [decode|encode|both|roundtrip]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from benchlib.measure import synthetic_inputs
from worldforge_amd import parallel
from worldforge_amd.vae import AutoencoderKLWan

dev = torch.device("cuda:0")
P, R = int(os.environ.get("P", "8")), int(os.environ.get("RANK_SIM", "4"))
what = sys.argv[1] if len(sys.argv) > 1 else "both"
z = torch.randn(1, 16, 21, 60, 104, device=dev)
video = torch.rand(1, 3, 81, 480, 832, device=dev) * 2 - 1
comm = parallel.LoopbackComm(P, R) if P > 1 else None
v = AutoencoderKLWan(dev, comm=comm, precision=os.environ.get("PREC", "fp16x3")).init_random(seed=1)
v.crop_to_mask = os.environ.get("CROP", "1") != "0"
_, ref, mask, _, _, _ = synthetic_inputs(81, 480, 832, dev)     # SURVEY 8d's mask: the hole grows to 35 % of the width
for it in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    if what in ("decode", "both"):
        v.decode(z, return_dict=False)
    if what in ("encode", "both"):
        v.encode(video)
    if what == "roundtrip":     # what an IRR injection runs: decode -> blend -> encode (row slabs never gathered on a sharded VAE)
        v.decode_blend_encode(z, ref, mask)
    e.record()
    torch.cuda.synchronize()
    print(f"iteration {it}: {what} {s.elapsed_time(e):.1f} ms", flush=True)
v.check_range()
