"""CPU: what the four runs of the UNMODIFIED reference recorded by tools/flf_reference_study.py say about the FLF gate at schedule length
(tests/golden/g19_flf_reference_{tdiff,farneback}_t{1,6}.npz: the job of tests/test_gpu_schedule_length.py through the imported reference
pipeline + scheduler + vendored AutoencoderKLWan + in-tree WanModel twin, torch.set_num_threads(1) and (6)).  These are the facts DESIGN
section 4b and the README quote; the GPU side of the same fixtures is tests/test_gpu_schedule_length.py."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g19(name):
    return os.path.join(ROOT, "tests", "golden", f"g19_flf_reference_{name}.npz")


def test_reference_recorded_runs_thread_count_stability():
    """CPU-side facts of the fixtures themselves (no GPU work): the evidence DESIGN section 4b / README quote."""
    import json
    import numpy as np
    lists = {n: json.loads(str(np.load(_g19(n))["flf_lists"])) for n in ("tdiff_t1", "tdiff_t6", "farneback_t1", "farneback_t6")}
    assert lists["tdiff_t1"] == lists["tdiff_t6"]                       # the reference's own branch: thread-count stable on this job
    fb1, fb6 = dict(map(tuple, ((s, tuple(c)) for s, c in lists["farneback_t1"]))), dict((s, tuple(c)) for s, c in lists["farneback_t6"])
    first = next(s for s in sorted(fb1) if fb1[s] != fb6[s])
    assert first == 13                                                   # Farneback gate: the reference flips against itself late in the job
    fr = {n: np.load(_g19(n))["frames"].astype(np.float32) for n in ("tdiff_t1", "tdiff_t6", "farneback_t1", "farneback_t6")}
    psnr = lambda a, b: 10 * np.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-12))  # noqa: E731
    assert psnr(fr["tdiff_t1"], fr["tdiff_t6"]) >= 45.0
    assert psnr(fr["farneback_t1"], fr["farneback_t6"]) <= 30.0


def test_fixture_job_matches_the_gpu_study_job():
    """The fixtures describe exactly the job tools/vae_precision_study.study() runs on the GPU (its `fixture=` argument asserts this too)."""
    import json
    import numpy as np
    for n, backend in (("tdiff_t1", "tdiff"), ("tdiff_t6", "tdiff"), ("farneback_t1", "farneback"), ("farneback_t6", "farneback")):
        z = np.load(_g19(n))
        job = json.loads(str(z["job"]))
        assert job == dict(dit="d1024 x 4 layers x 8 heads", frames=17, height=128, width=128, steps=20, guided_steps=15, round_trips=31,
                           flow_backend=backend)
        assert z["latents"].shape == (20, 1, 16, 5, 16, 16) and z["frames"].shape == (5, 128, 128, 3)
        assert len(json.loads(str(z["flf_lists"]))) == 15
