"""The rank launcher of `bench.py --gpus N` and the one-JSON-line plumbing.  Nothing in here initialises the GPU."""
from __future__ import annotations

import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _transport() -> str:
    """What the collectives of this run travel over: RCCL, or -- debug runs with every rank on one GPU -- gloo through the host."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return "gloo through the host: DEBUG transport, timings meaningless"
    return "RCCL"

FALLBACK_ARGS = ["--exchange", "gather", "--conservative"]   # attempt 2: one all-gather per layer, no CFG groups, whole-job halo all-gather
ATTEMPT_BUDGET_S = 900.0


def progress(msg: str):
    """Rank-0 progress line on stderr at a phase boundary (init, groups, calibration candidate k, warm-up step i, window start / end):
    what a first run on a real node leaves behind when it stops somewhere."""
    if os.environ.get("RANK", "0") == "0":
        t0 = float(os.environ.get("WF_BENCH_T0", "0") or 0)
        print(f"[bench +{time.time() - t0:7.1f}s] {msg}" if t0 else f"[bench] {msg}", file=sys.stderr, flush=True)


class _Coord:
    """What the supervisors of one job agree through: a c10d TCPStore (CPU only -- no GPU, no process group).  Under
    `torch.distributed.run` it is the agent's store (every torchrun worker is a client); self-launched, the parent hosts it.  The rank
    processes rendezvous through the SAME store behind a per-attempt prefix (worldforge_amd.parallel.init, WF_STORE_PREFIX), so a second
    attempt never meets the keys of the first and no second port has to be guessed."""

    def __init__(self, host: str, port: int, is_master: bool):
        from datetime import timedelta

        import torch.distributed as dist
        self.store = dist.TCPStore(host, port, None, is_master, timeout=timedelta(seconds=60), wait_for_workers=False)

    def fail(self, attempt: int, info: dict):
        if not self.failed(attempt):
            self.store.set(f"wfsup/a{attempt}/failed", json.dumps(info))

    def failed(self, attempt: int):
        return self.store.check([f"wfsup/a{attempt}/failed"])

    def failure(self, attempt: int) -> dict:
        return json.loads(self.store.get(f"wfsup/a{attempt}/failed").decode())

    def done(self, attempt: int, n: int = 1) -> int:
        return self.store.add(f"wfsup/a{attempt}/done", n)


def _tee(pipe, sink, keep: list, limit: int = 40):
    """Forward a child's stderr line by line and remember its last lines."""
    for line in iter(pipe.readline, b""):
        try:
            sink.write(line)
            sink.flush()
        except (OSError, ValueError):
            pass
        keep.append(line.decode(errors="replace").rstrip())
        del keep[:-limit]
    pipe.close()


def supervise(my_ranks, world: int, argv, script: str, coord: _Coord, host: str, port: int, budget_s: float = ATTEMPT_BUDGET_S,
              fallback_args=FALLBACK_ARGS, local_world: int = None) -> int:
    """Run the ranks `my_ranks` of a `world`-rank job as FRESH CHILD PROCESSES, at most twice: attempt 1 with `argv`; if ANY rank of the job
    exits non-zero or the attempt outlives its wall budget, every supervisor stops its children and all of them start attempt 2 with the
    conservative configuration (`fallback_args`), whose JSON line carries `"fallback": true` and the first attempt's last stderr lines.
    The supervisor never touches the GPU and never re-execs; rank 0's stdout (the one JSON line) is held back until its attempt is known
    to have succeeded everywhere.  -> exit code."""
    import signal

    my_ranks = list(my_ranks)
    live = []   # child processes that may still run: never left behind, whatever ends this supervisor (the finally below; SIGTERM from a
                # launcher that gives up becomes SystemExit so that it runs)

    def _on_term(signum, frame):
        raise SystemExit(128 + signum)

    old_handlers = {}
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            old_handlers[sig] = signal.signal(sig, _on_term)
        except (ValueError, OSError):   # not the main thread (tests): leave the handlers alone
            pass
    try:
        return _supervise_attempts(my_ranks, world, argv, script, coord, host, port, budget_s, fallback_args, local_world, live)
    finally:
        for p in live:
            if p.poll() is None:
                p.kill()
        for sig, h in old_handlers.items():
            signal.signal(sig, h)


def _supervise_attempts(my_ranks, world, argv, script, coord, host, port, budget_s, fallback_args, local_world, live) -> int:
    import subprocess
    import tempfile
    import threading

    info_path = None
    for attempt in (1, 2):
        args = list(argv) + (list(fallback_args) if attempt == 2 else [])
        procs, tails, outs, threads = [], {}, {}, []
        for r in my_ranks:
            env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_ELASTIC"))}
            env.update(RANK=str(r), LOCAL_RANK=str(r if local_world is None else r % local_world), WORLD_SIZE=str(world),
                       LOCAL_WORLD_SIZE=str(local_world or world), MASTER_ADDR=host, MASTER_PORT=str(port), WF_BENCH_CHILD="1",
                       WF_BENCH_ATTEMPT=str(attempt), WF_STORE_PREFIX=f"wfpg/a{attempt}", WF_BENCH_T0=str(time.time()))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if info_path:
                env["WF_BENCH_FALLBACK_INFO"] = info_path
            p = subprocess.Popen([sys.executable, script] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            live.append(p)
            tails[r], outs[r] = [], []
            for pipe, fn, a in ((p.stderr, _tee, (p.stderr, sys.stderr.buffer, tails[r])),
                                (p.stdout, lambda q, acc: acc.append(q.read()), (p.stdout, outs[r]))):
                th = threading.Thread(target=fn, args=a, daemon=True)
                th.start()
                threads.append(th)
            procs.append((r, p))
        t0 = time.time()
        pending = dict(procs)
        stopped, t_kill = False, float("inf")
        while pending:
            for r, p in list(pending.items()):
                rc = p.poll()
                if rc is None:
                    continue
                del pending[r]
                if rc != 0 and not stopped:
                    coord.fail(attempt, {"reason": f"rank {r} exited with code {rc}", "rank": r, "rc": rc, "stderr_tail": tails[r][-12:]})
            if pending and not stopped:
                over = time.time() - t0 > budget_s
                if over:   # nobody knows which rank the others are waiting for: the last lines of every rank still running
                    coord.fail(attempt, {"reason": f"wall budget of {budget_s:.0f} s exceeded (ranks {sorted(pending)} still running)",
                                         "rank": min(pending), "rc": None,
                                         "stderr_tail": [f"[rank {r}] {ln}" for r in sorted(pending) for ln in tails[r][-4:]][-24:]})
                if over or coord.failed(attempt):   # someone failed (here or under another supervisor): the others would wait for ever
                    stopped = True
                    for p in pending.values():
                        p.terminate()
                    t_kill = time.time() + 15
            if pending and stopped and time.time() > t_kill:
                for p in pending.values():
                    p.kill()
            time.sleep(0.1)
        for th in threads:
            th.join(timeout=5)
        # every supervisor's children are gone before anyone decides: a late failure elsewhere must turn THIS attempt into a failure too
        n = coord.done(attempt, len(my_ranks))
        t_wait = time.time() + 120
        while n < world and time.time() < t_wait:
            time.sleep(0.1)
            n = coord.done(attempt, 0)
        if not coord.failed(attempt) and n >= world:
            for r in my_ranks:
                data = b"".join(outs[r])
                if data:
                    os.write(1, data)
            return 0
        if attempt == 2:
            f = coord.failure(2) if coord.failed(2) else {"reason": "supervisors did not all report"}
            print(f"bench.py: the conservative second attempt failed too: {f.get('reason')}", file=sys.stderr)
            return int(f.get("rc") or 1)
        first = coord.failure(1) if coord.failed(1) else {"reason": "supervisors did not all report"}
        print(f"bench.py: attempt 1 failed ({first.get('reason')}); ONE relaunch in fresh processes with {' '.join(fallback_args)}", file=sys.stderr)
        fd, info_path = tempfile.mkstemp(prefix="wf_bench_fallback_", suffix=".json")
        with os.fdopen(fd, "w") as fh:
            json.dump(first, fh)
    return 1


def launch_ranks(n: int, argv, script: str = None, budget_s: float = ATTEMPT_BUDGET_S) -> int:
    """`python bench.py --gpus N` without a launcher: this parent hosts the rendezvous store on 127.0.0.1 and supervises N fresh rank
    processes (one per GPU) -- see supervise().  It never touches the GPU (no torch.cuda call that initialises HIP, no libwf_hip.so) and
    never re-execs.  (The reference's own multi-GPU entry has the same shape: run_upscale.py:71-77 reads RANK / LOCAL_RANK from a
    launcher.)"""
    import socket

    share = bool(os.environ.get("WF_SHARE_GPU"))
    have = visible_gpu_count()  # from the environment / sysfs: no torch.cuda call, nothing that could initialise HIP in the launcher
    if not share and have is not None and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible (WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo runs all ranks on one GPU "
              "as a debug configuration)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    coord = _Coord("127.0.0.1", port, is_master=True)
    return supervise(range(n), n, argv, script or os.path.join(ROOT, "bench.py"), coord, "127.0.0.1", port, budget_s)


def supervise_under_launcher(argv, script: str = None, budget_s: float = ATTEMPT_BUDGET_S) -> int:
    """A rank started by `python -m torch.distributed.run` (the driver's N > 1 command): become the supervisor of THIS rank -- the real rank
    runs as a child, so that a failure or a hang of the first attempt anywhere in the job still ends in one conservative relaunch in fresh
    processes.  The supervisors agree through the launcher's own rendezvous store."""
    host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    hosted = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
    coord = _Coord(host, port, is_master=(not hosted and rank == 0))
    return supervise([rank], world, argv, script or os.path.join(ROOT, "bench.py"), coord, host, port, budget_s,
                     local_world=int(os.environ.get("LOCAL_WORLD_SIZE", world)))


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT touching the HIP runtime: the *_VISIBLE_DEVICES lists if set, else the KFD topology
    (a node with simd_count > 0 is a GPU).  None if neither source is readable (the pre-check is then skipped: a rank that finds no
    device fails non-zero and the launcher propagates it)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout of every rank (seen with RCCL 2.26 on the
    GPU box, flushed at exit, i.e. AFTER the line), and other libraries may chat there too: keep a private duplicate of fd 1 for the JSON
    line and point fd 1 at stderr for everything else, in every rank, before anything is initialised."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(out: dict):
    info = os.environ.get("WF_BENCH_FALLBACK_INFO")
    if info:   # this is the conservative second attempt (supervise): say so, with what ended the first one
        try:
            with open(info) as f:
                out = dict(out, fallback=True, first_attempt=json.load(f))
        except (OSError, ValueError):
            out = dict(out, fallback=True)
    data = (json.dumps(out) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def shutdown_comm():
    """Tear the process group down before exit (RCCL otherwise warns about leaked resources; LoopbackComm has none)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def rank_env(a):
    """(rank, local_rank, world) from the launcher's environment; --gpus must agree with WORLD_SIZE."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher: it starts the ranks itself)")
    if os.environ.get("WF_SHARE_GPU"):  # debug: all ranks on one GPU (with WF_COMM_BACKEND=gloo) to exercise the N > 1 path
        local_rank = 0
    return rank, local_rank, world
