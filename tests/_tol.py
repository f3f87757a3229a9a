"""Tolerance bars that remember what they measured (VERDICT r3 "weak" #7: tests were 6-8x looser than the errors they see).

`within(name, measured, bound)` asserts `measured <= bound`; with `WF_TOL_LOG=<file>` it also appends `name measured bound` so that one
run of the suite on the GPU shows every bar next to the error it actually sees.  Policy: a bar sits at <= 2x the error measured on an
MI355X (the value is quoted in the comment at the call site); profiles/r5_tolerances.txt is the log of the round-end run at HEAD (tools/gpurun_scripts/final.sh regenerates it).
"""
import os


def within(name, measured, bound):
    measured, bound = float(measured), float(bound)
    path = os.environ.get("WF_TOL_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{name} measured {measured:.4e} bound {bound:.4e} ratio {bound / max(measured, 1e-30):.2f}\n")
    assert measured <= bound, f"{name}: measured {measured:.4e} > bound {bound:.4e}"
    return measured
