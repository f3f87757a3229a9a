#!/bin/bash
# round 4, b: the whole GPU suite with every tolerance bar logged next to the error it sees (tests/_tol.py, WF_TOL_LOG) + the contract bench line
#   -> gpurun_out/r4/b_tolerances.txt, b_pytest_gpu.log, b_bench.json
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/b_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r4/b_tolerances.txt python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 > gpurun_out/r4/b_pytest_gpu.log
python bench.py > gpurun_out/r4/b_bench.json 2> gpurun_out/r4/b_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/b_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4/b_smoke.log
tail -5 gpurun_out/r4/b_pytest_gpu.log; head -c 600 gpurun_out/r4/b_bench.json; echo; tail -2 gpurun_out/r4/b_smoke.log
