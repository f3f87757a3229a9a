"""GPU: the guided job at SCHEDULE LENGTH: 20-step schedule, 15 guided steps x 2 rounds = 31 VAE decode -> encode round
trips, FLF swapping channels on 9 of the 15 gates (the > 10 branch included), 17 frames of 128 x 128, d = 1024 x 4-layer DiT, against the
CPU oracle's recorded result for exactly this job (tests/golden/g17_schedule_length_oracle.npz, written by
`python tools/vae_precision_study.py --oracle-only --save-fixture ...`: frames (fp16), per-step latents, gate decisions, gate similarities).

What is asserted, and why in this form: the FLF gate (SCHED:408-437) is a DISCRETE decision on 16 similarities; where two of them are
nearly tied at the selection boundary, arithmetic differences of bf16-activation size (the DiT runs bf16 activations in the product and in
the reference's GPU path, fp32 in the oracle) flip the decision, and a flipped channel swap is an O(1) change of x0 -- no precision of
the VAE can prevent that.  So:
  * with the oracle's gate decisions replayed (scheduler.flf_replay) the frames must agree with the oracle: PSNR >= 40 dB for the
    default fp32-class VAE -- the arithmetic of all 31 round trips, 54 DiT forwards and the scheduler;
  * free-running, the product's decisions must equal the oracle's at every gate whose decision margin (distance of the similarities from
    a tie at the selection boundary, tools/vae_precision_study.decision_margin) exceeds MARGIN; flips are only tolerated below it;
  * the opt-in bf16-operand VAE is held to the same replayed-decision bar (its noise is what moves near-tied gates, not the frames)."""
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "g17_schedule_length_oracle.npz")
MARGIN = 5e-2   # similarities live in [0, 1]; the measured |delta sim| between two arithmetically close runs of this job (product vs oracle, bf16- vs fp32-class VAE) is 0.01-0.07 per gate: the uint8 quantisation inside the Farneback input (SCHED:175-176) turns bf16-level latent differences into different flows


@pytest.fixture(scope="module")
def study():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import vae_precision_study as vs
    return vs.study(fixture=FIX, verbose=False)


def test_replayed_decisions_frames_match_oracle(study):
    rep = study["decisions_replayed"]
    print({k: v for k, v in rep.items() if "psnr" in k})
    assert rep["decisions_from"] == "fixture"   # g17: the oracle's recorded run
    assert rep["psnr_fp32vae_vs_oracle_db"] >= 40.0
    assert rep["psnr_bf16vae_vs_oracle_db"] >= 40.0
    assert min(rep["latent_db_fp32vae_vs_oracle_per_step"]) >= 30.0   # no step of the trajectory drifts away


def test_free_running_decisions_match_oracle_outside_near_ties(study):
    fr = study["free_running"]
    assert fr["flf_gates"] == 15 and fr["flf_swapping_gates"] >= 8
    margins = dict(fr["gate_margin_oracle"])
    got, want = dict(map(tuple, ((s, tuple(c)) for s, c in fr["flf_lists_fp32vae"]))), dict((s, tuple(c)) for s, c in fr["flf_lists_oracle"])
    flipped = False
    for step in sorted(want):
        if flipped:
            break   # after a flip the two trajectories are different jobs; later gates are not comparable
        if got[step] != want[step]:
            m = margins.get(step)
            assert m is not None and m < MARGIN, (step, got[step], want[step], m)
            flipped = True
    print("first flip at", next((s for s in sorted(want) if got[s] != want[s]), None), "margins", margins)


# ---------------------------------------------------------------------------------------------------------------------------------------
# The same job recorded from the UNMODIFIED reference pipeline + scheduler + vendored AutoencoderKLWan + in-tree WanModel twin
# (tools/flf_reference_study.py, tests/golden/g19_flf_reference_*.npz), with torch.set_num_threads(1) and (6).  What those four runs say
# (DESIGN section 4b): on the temporal-difference branch (the one the reference executes here: no cv2) the reference's gate decisions are
# identical at 1 and 6 threads and its frames agree to 51.5 dB; with the Farneback gate (cv2 served by oracle/farneback.py) the reference
# DISAGREES WITH ITSELF between 1 and 6 threads at gates 13 and 14 and its frames end at 22.7 dB.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _g19(name):
    return os.path.join(ROOT, "tests", "golden", f"g19_flf_reference_{name}.npz")


def _study(fixture, backend):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import vae_precision_study as vs
    return vs.study(fixture=fixture, flow_backend=backend, verbose=False, skip_bf16=True)


def test_tdiff_branch_free_running_vs_reference_recorded_run():
    """The golden-pinned branch at schedule length, FREE-RUNNING, against the reference's own recorded run (6 threads): same gate decisions
    wherever the margin exceeds the similarity noise of the bf16 DiT path, frames >= 40 dB when no gate flipped, and >= 40 dB with the
    reference's decisions replayed in any case."""
    st = _study(_g19("tdiff_t6"), "tdiff")
    fr, rep = st["free_running"], st["decisions_replayed"]
    print({k: v for k, v in fr.items() if "psnr" in k or "same" in k}, {k: v for k, v in rep.items() if "psnr" in k})
    print("margins", fr["gate_margin_oracle"], "max |dsim|", fr["gate_max_sim_delta_fp32vae_vs_oracle"])
    assert rep["psnr_fp32vae_vs_oracle_db"] >= 40.0
    got, want = dict((s, tuple(c)) for s, c in fr["flf_lists_fp32vae"]), dict((s, tuple(c)) for s, c in fr["flf_lists_oracle"])
    margins = dict(fr["gate_margin_oracle"])
    noise = max(d for _, d in fr["gate_max_sim_delta_fp32vae_vs_oracle"] if d is not None)
    flips = [s for s in sorted(want) if got[s] != want[s]]
    if not flips:
        assert fr["psnr_fp32vae_vs_oracle_db"] >= 40.0                   # the north_star bar, free-running, against the reference itself
    else:
        m = margins.get(flips[0])
        assert m is not None and m <= 2 * noise, (flips[0], m, noise)   # a flip only where the decision was inside the measured noise


@pytest.mark.parametrize("threads", ["t6", "t1"])
def test_farneback_branch_replayed_vs_reference_recorded_runs(threads):
    """The deployed branch against BOTH of the reference's runs (which disagree with each other from gate 13 on): with the respective
    run's decisions replayed the product reproduces EACH of them to >= 40 dB -- the arithmetic of the path is not what separates them."""
    st = _study(_g19("farneback_" + threads), "farneback")
    rep, fr = st["decisions_replayed"], st["free_running"]
    print(threads, {k: v for k, v in rep.items() if "psnr" in k}, "free-running", fr["psnr_fp32vae_vs_oracle_db"], "same lists", fr["flf_same_fp32vae_vs_oracle"])
    assert rep["psnr_fp32vae_vs_oracle_db"] >= 40.0
