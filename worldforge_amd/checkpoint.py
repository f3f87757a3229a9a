"""Checkpoint reader for the diffusers directories the reference loads (INFER:179-197: `AutoencoderKLWan.from_pretrained(...,
subfolder="vae")`, `WanImageToVideoPipeline.from_pretrained(model_id, ...)` -> `transformer/`): safetensors files, single or sharded
with a `*.safetensors.index.json` weight map.

The safetensors container is parsed here directly (8-byte little-endian header length, a JSON header {name: {dtype, shape,
data_offsets}}, then the raw little-endian tensor bytes) and memory-mapped, so a 28 GB transformer checkpoint is never copied on the
host more than once per tensor on its way to HBM.  No dependency on the `safetensors` package (tests use it as the writer)."""
from __future__ import annotations

import json
import mmap
import os
import struct
from typing import Dict, Iterable, Optional

import numpy as np
import torch

_DTYPES = {
    "F64": (torch.float64, 8), "F32": (torch.float32, 4), "F16": (torch.float16, 2), "BF16": (torch.bfloat16, 2),
    "I64": (torch.int64, 8), "I32": (torch.int32, 4), "I16": (torch.int16, 2), "I8": (torch.int8, 1), "U8": (torch.uint8, 1),
    "BOOL": (torch.bool, 1),
}


def read_header(path: str):
    """-> (header dict without __metadata__, byte offset of the data section)."""
    with open(path, "rb") as f:
        raw = f.read(8)
        if len(raw) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
        (n,) = struct.unpack("<Q", raw)
        size = os.path.getsize(path)
        if n <= 0 or 8 + n > size:
            raise ValueError(f"{path}: not a safetensors file (header length {n} exceeds the file size {size})")
        hdr = json.loads(f.read(n).decode("utf-8"))
    hdr.pop("__metadata__", None)
    return hdr, 8 + n


def load_file(path: str, names: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
    """All (or the named) tensors of one safetensors file as CPU tensors backed by a private copy-on-write memory map."""
    hdr, base = read_header(path)
    want = set(hdr) if names is None else set(names)
    missing = want - set(hdr)
    if missing:
        raise KeyError(f"{path}: tensors not in the file: {sorted(missing)[:5]}")
    size = os.path.getsize(path)
    out: Dict[str, torch.Tensor] = {}
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)
    buf = np.frombuffer(mm, dtype=np.uint8)
    for name in sorted(want):
        e = hdr[name]
        if e["dtype"] not in _DTYPES:
            raise ValueError(f"{path}: tensor {name!r} has unsupported dtype {e['dtype']}")
        dt, isz = _DTYPES[e["dtype"]]
        lo, hi = e["data_offsets"]
        numel = int(np.prod(e["shape"])) if e["shape"] else 1
        if hi - lo != numel * isz or base + hi > size or lo < 0:
            raise ValueError(f"{path}: tensor {name!r} has inconsistent offsets {lo}:{hi} for shape {e['shape']} {e['dtype']}")
        t = torch.from_numpy(buf[base + lo:base + hi])
        out[name] = (t.view(dt) if numel else torch.empty(0, dtype=dt)).reshape(e["shape"])
    return out


def load_dir(folder: str) -> Dict[str, torch.Tensor]:
    """A diffusers component directory: `<name>.safetensors.index.json` + shards, or the single `*.safetensors` file."""
    if os.path.isfile(folder):
        return load_file(folder)
    files = sorted(os.listdir(folder))
    idx = [f for f in files if f.endswith(".safetensors.index.json")]
    if idx:
        if len(idx) > 1:
            raise ValueError(f"{folder}: several index files {idx}")
        with open(os.path.join(folder, idx[0])) as f:
            wm = json.load(f)["weight_map"]
        by_file: Dict[str, list] = {}
        for name, fn in wm.items():
            by_file.setdefault(fn, []).append(name)
        out: Dict[str, torch.Tensor] = {}
        for fn, names in sorted(by_file.items()):
            p = os.path.join(folder, fn)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{folder}: shard {fn} named by {idx[0]} is missing")
            out.update(load_file(p, names))
        return out
    st = [f for f in files if f.endswith(".safetensors")]
    if len(st) != 1:
        raise FileNotFoundError(f"{folder}: expected one *.safetensors file or an index, found {st}")
    return load_file(os.path.join(folder, st[0]))
