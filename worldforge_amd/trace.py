"""Tracing of the sampling loop (the reference has only tqdm + prints; the build's counterpart is roctx ranges
per (step, round, phase) for rocprofv3 --marker-trace, plus a JSON timing log).

    tr = Tracer()                      # or Tracer.from_env(): enabled by WF_TRACE=<path.json>
    with tr.range("dit_cfg_pair", step=i, round=r): ...
    tr.finish("timing.json")           # resolves the HIP events (one sync, at the end) and writes the log

A range pushes / pops a roctx range (torch.cuda.nvtx is roctx on ROCm) and records a HIP event pair on the current stream; nothing
synchronises inside the loop.  Disabled tracers cost one attribute test per range."""
from __future__ import annotations

import contextlib
import json
import os
from typing import Dict, List, Optional

import torch


class Tracer:
    def __init__(self, enabled: bool = True, roctx: bool = True):
        self.enabled = enabled and torch.cuda.is_available()
        self.roctx = roctx
        self._open: List = []
        self.records: List[Dict] = []
        self._pending: List = []

    @classmethod
    def from_env(cls) -> "Tracer":
        path = os.environ.get("WF_TRACE")
        t = cls(enabled=bool(path))
        t.path = path
        return t

    def reset(self):
        """Forget the records of earlier calls (open ranges of an aborted call included)."""
        self.records = []
        self._pending = []

    @contextlib.contextmanager
    def range(self, name: str, **tags):
        if not self.enabled:
            yield
            return
        label = name + "".join(f" {k}={v}" for k, v in tags.items())
        if self.roctx:
            torch.cuda.nvtx.range_push(label)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            yield
        finally:
            e1.record()
            if self.roctx:
                torch.cuda.nvtx.range_pop()
            self._pending.append((name, tags, e0, e1))

    def resolve(self) -> List[Dict]:
        """Turn the pending event pairs into {name, tags..., ms} records (synchronises once)."""
        if self._pending:
            torch.cuda.synchronize()
            for name, tags, e0, e1 in self._pending:
                self.records.append({"name": name, **tags, "ms": e0.elapsed_time(e1)})
            self._pending = []
        return self.records

    def summary(self) -> Dict[str, Dict[str, float]]:
        """{phase: {count, total_ms, mean_ms}} over the resolved records."""
        out: Dict[str, Dict[str, float]] = {}
        for r in self.resolve():
            s = out.setdefault(r["name"], {"count": 0, "total_ms": 0.0})
            s["count"] += 1
            s["total_ms"] += r["ms"]
        for s in out.values():
            s["mean_ms"] = s["total_ms"] / s["count"]
        return out

    def finish(self, path: Optional[str] = None):
        path = path or getattr(self, "path", None)
        recs = self.resolve()
        if path:
            with open(path, "w") as f:
                json.dump({"records": recs, "summary": self.summary()}, f, indent=1)
        return recs


NULL = Tracer(enabled=False)
