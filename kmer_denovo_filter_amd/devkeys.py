"""Key sets that stay in HBM between the discovery stages.

The reference hands k-mer sets from stage to stage as FASTA files
(``child_candidates.fa`` -> ``child_non_ref_kmers.fa`` -> ``after_mother.fa`` ->
``proband_unique.fa``; discovery/pipeline.py:207-226,286-304,515-532).  The
mirror keeps that contract -- every file is still written, in the reference's
``>{i}\\n{KMER}\\n`` form -- but a stage that finds the set of its input path in
this registry takes the keys from device memory (``kdf_load_filter_dev`` /
``kdf_query_dev``) instead of parsing the text back and copying it up again.
torch is the holder of the device arrays here, nothing more.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

_registry: Dict[str, Tuple[object, Optional[object], int, Tuple[int, int]]] = {}


def _key(path: str) -> str:
    return os.path.abspath(path)


def _stamp(path: str):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns)


def register(path: str, lo, hi, k: int):
    """``lo`` / ``hi``: int64 CUDA tensors (bit patterns of the uint64 key words); hi None for k <= 32.
    Call right after the FASTA at ``path`` was written: its size and mtime are the set's identity."""
    _registry[_key(path)] = (lo, hi, int(k), _stamp(path))


def lookup(path: str, k: int):
    """(lo, hi) device tensors registered for ``path`` at this k, or None.  A file that was rewritten since
    (a resumed run, an edited FASTA, a second trio in the same tmpdir) no longer matches and is parsed instead."""
    ent = _registry.get(_key(path))
    if ent is None:
        return None
    try:
        same = ent[2] == int(k) and _stamp(path) == ent[3]
    except OSError:
        same = False
    if not same:
        _registry.pop(_key(path), None)
        return None
    return ent[0], ent[1]


def forget(path: str):
    _registry.pop(_key(path), None)


def to_host(lo, hi) -> Tuple[np.ndarray, np.ndarray]:
    """Device key tensors -> the (lo, hi) uint64 arrays the FASTA writer takes."""
    hlo = lo.cpu().numpy().view(np.uint64)
    hhi = hi.cpu().numpy().view(np.uint64) if hi is not None else np.zeros(len(hlo), np.uint64)
    return hlo, hhi


def from_host(lo: np.ndarray, hi: Optional[np.ndarray], wide: bool, device: int = 0):
    import torch
    dev = torch.device("cuda", device)
    tlo = torch.from_numpy(np.ascontiguousarray(lo, dtype=np.uint64).view(np.int64)).to(dev)
    thi = torch.from_numpy(np.ascontiguousarray(hi, dtype=np.uint64).view(np.int64)).to(dev) if wide else None
    return tlo, thi


def dump_ge(eng, min_count: int, device: int = 0):
    """``jellyfish dump -c -L min_count`` into device tensors, ascending key order (the reference does not rely on
    the order, but a sorted dump makes the contract FASTA files byte-reproducible from run to run)."""
    import torch
    dev = torch.device("cuda", device)
    n = eng.count_ge(min_count)
    lo = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    hi = torch.empty(max(n, 1), dtype=torch.int64, device=dev) if eng.wide else None
    cnt = torch.empty(max(n, 1), dtype=torch.int32, device=dev)      # (the device sort carries the counts along)
    torch.cuda.current_stream(dev).synchronize()
    got = eng.export_ge_dev(min_count, lo.data_ptr(), hi.data_ptr() if hi is not None else None, cnt.data_ptr(), n,
                            sorted_=True) if n else 0
    if got != n:
        raise RuntimeError(f"dump -L {min_count}: {got} entries written, {n} counted")
    return lo[:n], (hi[:n] if hi is not None else None)


def query(eng, lo, hi, device: int = 0):
    """``jellyfish query``: uint32 counts (as int64 values) of the keys, input order, on the device."""
    import torch
    dev = torch.device("cuda", device)
    out = torch.zeros(lo.numel(), dtype=torch.int32, device=dev)
    if lo.numel():
        torch.cuda.current_stream(dev).synchronize()
        eng.query_dev(lo.data_ptr(), hi.data_ptr() if hi is not None else None, lo.numel(), out.data_ptr())
        eng.synchronize()
    return out.to(torch.int64) & 0xFFFFFFFF
