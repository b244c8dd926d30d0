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


def _index_layout(path: str, expect_k: Optional[int]):
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
    n = (os.path.getsize(path) - off) // rec.itemsize
    return k, kb, cb, rec, off, n


def index_records(path: str) -> int:
    """Number of records of an index (from the file size; nothing is read)."""
    return _index_layout(path, None)[5]


def _decode(data, kb: int, cb: int):
    """One block of records -> (lo, hi or None, counts uint32).  Little-endian byte strings of any length up to 16 / 8
    bytes; the usual widths (8-byte keys, 4-byte counters) are plain views of one contiguous copy per field."""
    n = len(data)
    kraw = np.ascontiguousarray(data["k"])
    if kb == 8:
        lo, hi = kraw.view("<u8").reshape(n), None
    else:
        kbytes = np.zeros((n, 16), dtype=np.uint8)
        kbytes[:, :kb] = kraw
        lo = np.ascontiguousarray(kbytes[:, :8]).view("<u8").reshape(n)
        hi = np.ascontiguousarray(kbytes[:, 8:]).view("<u8").reshape(n) if kb > 8 else None
    craw = np.ascontiguousarray(data["c"])
    if cb == 4:
        counts = craw.view("<u4").reshape(n)
    else:
        cbytes = np.zeros((n, 8), dtype=np.uint8)
        cbytes[:, :min(cb, 8)] = craw[:, :8]
        c64 = cbytes.view("<u8").reshape(n)
        if cb > 8:
            c64 = np.where(craw[:, 8:].any(axis=1), np.uint64(0xFFFFFFFFFFFFFFFF), c64)
        counts = np.minimum(c64, np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return lo, hi, counts


def iter_index(path: str, expect_k: Optional[int] = None, chunk_records: int = 1 << 24, part: int = 0, parts: int = 1):
    """Yield (k, lo, hi or None, counts) blocks of at most ``chunk_records`` records.  The file is memory-mapped and
    decoded block by block, so a whole-genome index (2.5e9 records, 30 GB on disk) needs ~0.5 GB of host memory at a
    time where `read_index` would need every record decoded at once (Jellyfish itself mmaps the file,
    reference discovery/pipeline.py:286-288)."""
    k, kb, cb, rec, off, n = _index_layout(path, expect_k)
    if n == 0:
        return
    mm = np.memmap(path, dtype=rec, mode="r", offset=off, shape=(n,))
    first, last = n * part // parts, n * (part + 1) // parts      # this part's records (one rank's share of the index)
    try:
        for a in range(first, last, chunk_records):
            lo, hi, counts = _decode(mm[a:min(a + chunk_records, last)], kb, cb)
            yield k, lo, hi, counts
    finally:
        del mm


def load_index_into(engine, path: str, expect_k: Optional[int] = None, chunk_records: int = 1 << 24,
                    part: int = 0, parts: int = 1) -> int:
    """Stream an index into ``engine``'s table (`jellyfish query`'s view of the .jf): the table is sized once for the
    record count, then the blocks are added one by one.  ``part`` of ``parts``: only that share of the records (the
    index sharded over the ranks of a multi-GPU job).  Returns the number of records of the whole index."""
    k, _, _, _, _, n = _index_layout(path, expect_k)
    if k != engine.k:
        raise ValueError(f"{path}: index has k={k}, the engine counts k={engine.k}")
    if n:
        engine.reserve(max(1, n // parts + 1))
    for _, lo, hi, counts in iter_index(path, expect_k, chunk_records, part, parts):
        engine.add_pairs(lo, hi, counts)
    return n


def read_index(path: str, expect_k: Optional[int] = None):
    """-> (k, lo, hi, counts) ; hi is all zero for k <= 32.  Counts saturate at 2^32-1.  Whole file in memory:
    callers that only feed an engine use `load_index_into`."""
    k = _index_layout(path, expect_k)[0]
    los, his, cnts = [], [], []
    for _, lo, hi, counts in iter_index(path, expect_k):
        los.append(lo); his.append(hi if hi is not None else np.zeros(len(lo), np.uint64)); cnts.append(counts)
    if not los:
        return k, np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    return k, np.concatenate(los), np.concatenate(his), np.concatenate(cnts)


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
