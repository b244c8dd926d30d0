"""On-disk k-mer indexes (the ``.jf`` files the reference passes between stages).

Reader: real Jellyfish ``binary/sorted`` files -- the format of ``--ref-jf`` /
``{ref}.k{k}.jf`` (core/jellyfish_wrappers.py:299-304) -- and this package's own
``kdf/sorted`` files.  Both share the container Jellyfish uses (SURVEY.md
section 0.5): 9 ASCII digits = byte length of the JSON header including its NUL
padding, the JSON header, then fixed records of ceil(key_len/8) key bytes (LE)
+ counter_len count bytes (LE).

Writers: ``kdf/sorted`` (records in ascending key order: ``write_index``, what
the mirrors write by default) and Jellyfish's own ``binary/sorted``
(``write_jellyfish_index``: records in the order of their hash position under the
header's GF(2) matrix -- position = XOR of ``matrix1.columns[i]`` over the set bits
``key_len - 1 - i`` of the key, ties in ascending key order).  That rule is pinned
by the reference's real Jellyfish file (tests/golden/giab/mini_ref.fa.k31.jf:
its 45 275 records are in exactly that order, and re-writing them from a shuffled
copy under its matrix gives the same bytes, tests/test_jf_helpers.py); that a real
``jellyfish query`` accepts a header generated HERE is unverified -- there is no
Jellyfish in this image.  ``KDF_JF_FORMAT=jellyfish``
makes the mirrors write ``{ref}.k{k}.jf`` that way.
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


# ---------------------------------------------------------------------------
# Jellyfish's own binary/sorted files
def jf_reprobes(max_reprobe: int = 126):
    """The reprobe offsets a Jellyfish 2 header lists (the fixture's: 1, then the triangular numbers)."""
    return [1] + [i * (i + 1) // 2 for i in range(1, max_reprobe + 1)]


def _gf2_rank(cols, r: int) -> int:
    rows = [int(c) for c in cols]                       # (column vectors of r bits: rank of the set)
    rank = 0
    for bit in range(r):
        piv = next((j for j in range(rank, len(rows)) if (rows[j] >> bit) & 1), None)
        if piv is None:
            continue
        rows[rank], rows[piv] = rows[piv], rows[rank]
        for j in range(len(rows)):
            if j != rank and (rows[j] >> bit) & 1:
                rows[j] ^= rows[rank]
        rank += 1
    return rank


def jf_make_matrix(key_len: int, r: int, seed: int = 0x6b6466):
    """A pseudo-random r x key_len GF(2) matrix of full row rank, as `key_len` column words (deterministic: the same
    (key_len, r) always gives the same file)."""
    rng = np.random.default_rng([seed, key_len, r])
    while True:
        cols = [int(x) & ((1 << r) - 1) for x in rng.integers(0, 1 << 63, key_len, dtype=np.uint64)]
        if _gf2_rank(cols, r) == min(r, key_len):
            return cols


def jf_positions(columns, key_len: int, lo: np.ndarray, hi: Optional[np.ndarray] = None) -> np.ndarray:
    """Hash position of every key under a header's matrix1: XOR of columns[i] over the set bits key_len - 1 - i of the key
    (bit 0 = the last base's low bit; bits >= 64 live in `hi`).  The bit order is the one under which the reference's
    real Jellyfish file is sorted; for key_len > 64 it is the natural extension and pinned by nothing."""
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    out = np.zeros(len(lo), dtype=np.uint64)
    one = np.uint64(1)
    for i, col in enumerate(columns):
        b = key_len - 1 - i
        if b < 0:
            break
        word = lo if b < 64 else np.ascontiguousarray(hi, dtype=np.uint64)
        bit = (word >> np.uint64(b & 63)) & one
        out ^= (np.uint64(0) - bit) & np.uint64(col)
    return out


def write_jellyfish_index(path: str, k: int, lo: np.ndarray, hi: Optional[np.ndarray], counts: np.ndarray,
                          size: Optional[int] = None, matrix_columns=None, cmdline=None, header: Optional[dict] = None) -> str:
    """Write a Jellyfish ``binary/sorted`` index of (key, count) records given in ANY order.
    `size` (a power of two; default: the first >= 2 n, at least 2^10) and `matrix_columns` describe the hash the
    records are ordered by; `header` (a dict parsed from another Jellyfish file) supplies both, and every other field,
    unchanged -- re-writing a real Jellyfish file from its own records and header gives its bytes back."""
    n = len(lo)
    key_len = 2 * k
    kb = (key_len + 7) // 8
    if header is not None:
        hd = dict(header)
        if int(hd["key_len"]) != key_len:
            raise ValueError(f"header is for key_len {hd['key_len']}, the keys have {key_len}")
        size = int(hd["size"]); matrix_columns = list(hd["matrix1"]["columns"])
    else:
        if size is None:
            size = 1 << 10
            while size < 2 * n:
                size <<= 1
        if size & (size - 1):
            raise ValueError("size must be a power of two")
        r = size.bit_length() - 1
        if matrix_columns is None:
            matrix_columns = jf_make_matrix(key_len, r)
        hd = {
            "alignment": 8, "canonical": True, "cmdline": list(cmdline or []), "counter_len": 4,
            "exe_path": "kmer_denovo_filter_amd (libkdf.so)", "format": JF_FORMAT, "key_len": key_len,
            "matrix1": {"c": key_len, "columns": [int(c) for c in matrix_columns], "identity": False, "r": r},
            "max_reprobe": 126, "reprobes": jf_reprobes(126), "size": int(size), "val_len": 7,
        }
    if int(hd.get("counter_len", 4)) != 4:
        raise ValueError("only 4-byte counters are written")
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    wide = k > 32
    hi = np.ascontiguousarray(hi, dtype=np.uint64) if wide else None
    pos = jf_positions(matrix_columns, key_len, lo, hi) & np.uint64(size - 1)
    order = np.lexsort((lo, hi, pos)) if wide else np.lexsort((lo, pos))     # position, then key
    body = json.dumps(hd).encode()
    body += b"\0" * ((-(9 + len(body))) % 8)
    rec = np.dtype([("k", "u1", (kb,)), ("c", "<u4")])
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(b"%09d" % len(body))
        fh.write(body)
        step = 1 << 22
        for a in range(0, n, step):                         # (records are assembled a slice at a time)
            o = order[a:a + step]
            data = np.zeros(len(o), dtype=rec)
            kbytes = np.zeros((len(o), 16), dtype=np.uint8)
            kbytes[:, :8] = lo[o].astype("<u8").view(np.uint8).reshape(len(o), 8)
            if wide:
                kbytes[:, 8:] = hi[o].astype("<u8").view(np.uint8).reshape(len(o), 8)
            data["k"] = kbytes[:, :kb]
            data["c"] = np.asarray(counts, dtype=np.uint32)[o]
            data.tofile(fh)
    os.replace(tmp, path)
    return path


def write_index_auto(path: str, k: int, lo, hi, counts, cmdline=None) -> str:
    """What the mirrors call for a user-visible index: ``kdf/sorted`` unless KDF_JF_FORMAT=jellyfish."""
    if os.environ.get("KDF_JF_FORMAT", "").lower() in ("jellyfish", "binary/sorted", "jf"):
        return write_jellyfish_index(path, k, lo, hi, counts, cmdline=cmdline)
    return write_index(path, k, lo, hi, counts, cmdline=cmdline)
