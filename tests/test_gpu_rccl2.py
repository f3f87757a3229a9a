"""GPU: the N > 1 path in REAL separate processes: Comm.all_gather_async on the communication stream, the token-sharded DiT
with its lock-step CFG pair and the row-sharded VAE, two ranks, each compared with the single-rank result (tests/rank_worker.py).
With >= 2 GPUs the ranks talk RCCL over xGMI (one GPU each); on a one-GPU box both ranks share GPU 0 and talk gloo -- the same code path
except for the transport (what `WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo python bench.py --gpus 2` does)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, env_extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env_extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r}/{world} ok" in o
    return outs


def test_two_ranks_in_separate_processes():
    if torch.cuda.device_count() >= 2:
        outs = _run(2, {})
        assert "nccl" in outs[0]
    else:
        outs = _run(2, {"WF_SHARE_GPU": "1", "WF_COMM_BACKEND": "gloo"})
        assert "gloo" in outs[0]


def test_one_rank_over_rccl():
    """A one-rank RCCL group on one GPU: the transport is trivial, but every call the N > 1 path makes (init_process_group("nccl",
    device_id), all_gather_into_tensor of bf16 / fp32 views on the communication stream, events, barrier) goes through RCCL itself,
    which the shared-GPU gloo run cannot show on a one-GPU box."""
    outs = _run(1, {})
    assert "nccl" in outs[0]


@pytest.mark.skipif(torch.cuda.device_count() < 4, reason="needs 4 GPUs")
def test_four_ranks_rccl():
    _run(4, {})
