"""The reference's intermediate k-mer file format ``>{i}\\n{KMER}\\n``
(utils.py:150-170, core/bam_scanner.py:34-46; written inline at
discovery/pipeline.py:221-226,297-304,523-532,573-583), vectorised, plus a
binary sidecar so that stages running on the engine skip the text round trip.

The FASTA stays the contract (callers and users may read it); the sidecar
``<path>.kdfkeys.npz`` is only trusted when it records the FASTA's exact size and
mtime.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

from .reads import _DEC, _ENC

_CHUNK = 1 << 22


def _sidecar(path: str) -> str:
    return path + ".kdfkeys.npz"


def _decode_matrix(lo: np.ndarray, hi: Optional[np.ndarray], k: int) -> np.ndarray:
    n = len(lo)
    out = np.empty((n, k), dtype=np.uint8)
    for i in range(k):
        sh = 2 * (k - 1 - i)
        src = (hi >> np.uint64(sh - 64)) if sh >= 64 else (lo >> np.uint64(sh))
        out[:, i] = _DEC[(src & np.uint64(3)).astype(np.intp)]
    return out


def write_kmer_fasta(path: str, lo: np.ndarray, hi: Optional[np.ndarray], k: int,
                     sidecar: bool = True) -> int:
    """Write keys as ``>{i}\\n{KMER}\\n`` (i from 0).  Returns the number written."""
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    n = len(lo)
    hi = np.zeros(n, np.uint64) if hi is None else np.ascontiguousarray(hi, dtype=np.uint64)
    with open(path, "wb") as fh:
        for a in range(0, n, _CHUNK):
            b = min(n, a + _CHUNK)
            seqs = _decode_matrix(lo[a:b], hi[a:b], k)
            idx = np.arange(a, b, dtype=np.int64)
            ndig = np.ones(b - a, dtype=np.int64)
            t = idx // 10
            while t.any():
                ndig += (t > 0)
                t //= 10
            for d in np.unique(ndig):
                sel = np.flatnonzero(ndig == d)          # contiguous run: indices are increasing
                m = len(sel)
                line = np.empty((m, 1 + d + 1 + k + 1), dtype=np.uint8)
                line[:, 0] = ord(">")
                v = idx[sel].copy()
                for j in range(d - 1, -1, -1):
                    line[:, 1 + j] = (v % 10) + ord("0")
                    v //= 10
                line[:, 1 + d] = 10
                line[:, 2 + d:2 + d + k] = seqs[sel]
                line[:, -1] = 10
                fh.write(line.tobytes())
    if sidecar:
        st = os.stat(path)
        np.savez(_sidecar(path), lo=lo, hi=hi, k=np.int64(k), size=np.int64(st.st_size),
                 mtime_ns=np.int64(st.st_mtime_ns))
    return n


def read_kmer_fasta_keys(path: str, k: int, canonical: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """Sequence lines of a k-mer FASTA -> (lo, hi) keys, file order.

    With ``canonical`` the keys are canonicalised, as Jellyfish does with an
    ``--if`` / ``query -s`` file under ``-C``.
    """
    sc = _sidecar(path)
    if os.path.exists(sc):
        try:
            z = np.load(sc)
            st = os.stat(path)
            if int(z["k"]) == k and int(z["size"]) == st.st_size and int(z["mtime_ns"]) == st.st_mtime_ns:
                return z["lo"], z["hi"]
        except Exception:  # noqa: BLE001  (stale or unreadable sidecar: fall back to the text)
            pass
    data = np.fromfile(path, dtype=np.uint8)
    if data.size == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    if data[-1] != 10:
        data = np.append(data, np.uint8(10))
    nl = np.flatnonzero(data == 10)
    starts = np.empty_like(nl)
    starts[0] = 0
    starts[1:] = nl[:-1] + 1
    lens = nl - starts
    # strip one trailing \r if present
    cr = (lens > 0) & (data[np.maximum(nl - 1, 0)] == 13)
    lens = lens - cr
    is_seq = (lens > 0) & (data[starts] != ord(">"))
    s, ln = starts[is_seq], lens[is_seq]
    if len(s) == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    if not (ln == k).all():
        raise ValueError(f"{path}: sequence line of length != k={k}")
    codes = _ENC[data[s[:, None] + np.arange(k)[None, :]]]
    if (codes > 3).any():
        raise ValueError(f"{path}: non-ACGT base in k-mer FASTA")
    codes = codes.astype(np.uint64)

    def pack(c):
        plo = np.zeros(len(c), np.uint64)
        phi = np.zeros(len(c), np.uint64)
        for i in range(k):
            sh = 2 * (k - 1 - i)
            if sh >= 64:
                phi |= c[:, i] << np.uint64(sh - 64)
            else:
                plo |= c[:, i] << np.uint64(sh)
        return plo, phi

    flo, fhi = pack(codes)
    if not canonical:
        return flo, fhi
    rlo, rhi = pack((np.uint64(3) - codes)[:, ::-1])
    fw = (fhi < rhi) | ((fhi == rhi) & (flo <= rlo))
    return np.where(fw, flo, rlo), np.where(fw, fhi, rhi)


def remove_with_sidecar(path: str):
    for p in (path, _sidecar(path)):
        if os.path.exists(p):
            os.remove(p)
