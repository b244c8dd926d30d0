"""On-disk k-mer indexes (the ``.jf`` files the reference passes between stages).

Reader: real Jellyfish ``binary/sorted`` files -- the format of ``--ref-jf`` /
``{ref}.k{k}.jf`` (core/jellyfish_wrappers.py:299-304) -- and this package's own
``kdf/sorted`` files.  Both share the container Jellyfish uses (SURVEY.md
section 0.5): 9 ASCII digits = byte length of the JSON header including its NUL
padding, the JSON header, then fixed records of ceil(key_len/8) key bytes (LE)
+ counter_len count bytes (LE).

Writer: ``kdf/sorted`` only (records in ascending key order).  Writing a file
that a real ``jellyfish query`` accepts needs Jellyfish's GF(2) matrix record
order; the reference never hands our files to Jellyfish once the engine replaces
it, so that is out of scope.
"""
from __future__ import annotations

import json
import os
from typing import Optional, Tuple

import numpy as np

KDF_FORMAT = "kdf/sorted"
JF_FORMAT = "binary/sorted"


def read_header(path: str) -> Tuple[dict, int]:
    with open(path, "rb") as fh:
        head = fh.read(9)
        if len(head) < 9 or not head.isdigit():
            raise ValueError(f"{path}: not a Jellyfish/kdf index (bad length prefix)")
        hlen = int(head)
        raw = fh.read(hlen)
    if len(raw) < hlen:
        raise ValueError(f"{path}: truncated index header")
    header = json.loads(raw.rstrip(b"\0").decode())
    return header, 9 + hlen


def read_index(path: str, expect_k: Optional[int] = None):
    """-> (k, lo, hi, counts) ; hi is all zero for k <= 32.  Counts saturate at 2^32-1."""
    header, off = read_header(path)
    fmt = header.get("format")
    if fmt not in (JF_FORMAT, KDF_FORMAT):
        raise ValueError(f"{path}: unsupported index format {fmt!r} (only {JF_FORMAT!r} and {KDF_FORMAT!r})")
    if fmt == JF_FORMAT and not header.get("canonical", False):
        raise ValueError(f"{path}: Jellyfish index was not built with -C (canonical)")
    key_len = int(header["key_len"])
    if key_len % 2:
        raise ValueError(f"{path}: odd key_len {key_len}")
    k = key_len // 2
    if expect_k is not None and k != expect_k:
        raise ValueError(f"{path}: index has k={k}, expected k={expect_k}")
    if k > 64:
        raise ValueError(f"{path}: k={k} is beyond the engine's key width")
    kb, cb = (key_len + 7) // 8, int(header["counter_len"])
    rec = np.dtype([("k", "u1", (kb,)), ("c", "u1", (cb,))])
    size = os.path.getsize(path) - off
    n = size // rec.itemsize
    data = np.fromfile(path, dtype=rec, count=n, offset=off)
    kbytes = np.zeros((n, 16), dtype=np.uint8)
    kbytes[:, :kb] = data["k"]
    lo = kbytes[:, :8].copy().view("<u8").reshape(n)
    hi = kbytes[:, 8:].copy().view("<u8").reshape(n)
    cbytes = np.zeros((n, 8), dtype=np.uint8)
    cbytes[:, :min(cb, 8)] = data["c"][:, :8]
    c64 = cbytes.view("<u8").reshape(n)
    counts = np.minimum(c64, np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return k, lo, hi, counts


def write_index(path: str, k: int, lo: np.ndarray, hi: Optional[np.ndarray], counts: np.ndarray,
                cmdline=None) -> str:
    """Write a ``kdf/sorted`` index (keys must already be in ascending order)."""
    n = len(lo)
    key_len = 2 * k
    kb = (key_len + 7) // 8
    header = {
        "alignment": 8, "canonical": True, "cmdline": list(cmdline or []), "counter_len": 4,
        "format": KDF_FORMAT, "key_len": key_len, "size": int(n),
        "exe_path": "kmer_denovo_filter_amd (libkdf.so)",
    }
    body = json.dumps(header).encode()
    total = 9 + len(body)
    pad = (-total) % 8
    body += b"\0" * pad
    rec = np.dtype([("k", "u1", (kb,)), ("c", "<u4")])
    data = np.zeros(n, dtype=rec)
    kbytes = np.zeros((n, 16), dtype=np.uint8)
    kbytes[:, :8] = np.ascontiguousarray(lo, dtype="<u8").view(np.uint8).reshape(n, 8)
    if hi is not None and k > 32:
        kbytes[:, 8:] = np.ascontiguousarray(hi, dtype="<u8").view(np.uint8).reshape(n, 8)
    data["k"] = kbytes[:, :kb]
    data["c"] = np.asarray(counts, dtype=np.uint32)
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(b"%09d" % len(body))
        fh.write(body)
        data.tofile(fh)
    os.replace(tmp, path)
    return path
