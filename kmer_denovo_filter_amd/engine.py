"""KmerEngine: Python face of one libkdf engine handle (one GPU, one table).

The methods map one-to-one onto the Jellyfish sub-commands the reference
shells out to ("External op" table):

    count(stream)            jellyfish count -m k -C            (insert mode)
    load_filter(keys)        --if filter.fa
    count_filtered(stream)   jellyfish count -m k -C --if ...
    export_ge(n)             jellyfish dump -c -L n             (ascending keys)
    query(keys)              jellyfish query idx -s kmers.fa    (input order)
    scan(stream)             JellyfishKmerQuery / Module-3 probe

No CPU fallback: constructing an engine without libkdf.so or without a GPU
raises.
"""
from __future__ import annotations

from ctypes import byref, c_uint64, c_void_p
from typing import Optional, Tuple

import numpy as np

from . import _native
from .reads import ReadStream, stream_words


def _vp(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class KmerEngine:
    MAX_K = 63

    def __init__(self, k: int, capacity_hint: int = 1 << 20, device: int = 0):
        if not (1 <= int(k) <= self.MAX_K):
            raise ValueError(f"k={k} outside the engine's range 1..{self.MAX_K}")
        self._lib = _native.load()
        self.k = int(k)
        self.wide = self.k > 32
        self.device = int(device)
        h = c_void_p()
        rc = self._lib.kdf_create(self.device, self.k, int(capacity_hint), byref(h))
        _native.check(rc, None)
        self._h = h

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.kdf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        _native.check(rc, self._h)

    def set_stream(self, hip_stream: Optional[int]):
        """Launch on an external hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        self._ck(self._lib.kdf_set_stream(self._h, c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._ck(self._lib.kdf_synchronize(self._h))

    def flush(self):
        """Apply everything the count calls have deferred (pending stream, partitioned passes) to the table now."""
        self._ck(self._lib.kdf_flush(self._h))
        return self

    def clear(self):
        self._ck(self._lib.kdf_clear(self._h))

    def reserve(self, n_keys: int):
        self._ck(self._lib.kdf_reserve(self._h, int(n_keys)))

    def stats(self) -> Tuple[int, int, int]:
        """(capacity slots, distinct keys, valid windows counted since clear)."""
        c, d, w = c_uint64(0), c_uint64(0), c_uint64(0)
        self._ck(self._lib.kdf_stats(self._h, byref(c), byref(d), byref(w)))
        return c.value, d.value, w.value

    def set_option(self, name: str, value: int):
        self._ck(self._lib.kdf_set_option(self._h, name.encode(), int(value)))

    def get_stat(self, name: str) -> int:
        from ctypes import c_int64
        v = c_int64(0)
        self._ck(self._lib.kdf_get_stat(self._h, name.encode(), byref(v)))
        return v.value

    def profile(self, enable: bool = True):
        self._ck(self._lib.kdf_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """(kernel ms, launches, stream positions) of the stream kernel since profile(True)."""
        import ctypes
        ms, n, p = ctypes.c_double(0), c_uint64(0), c_uint64(0)
        self._ck(self._lib.kdf_profile_read(self._h, byref(ms), byref(n), byref(p)))
        return ms.value, n.value, p.value

    def profile_stages(self):
        """([A0 hist+scans, A1 scatter, B finesort, C bucket] summed ms, binned passes)."""
        import ctypes
        ms = (ctypes.c_double * 4)()
        n = c_uint64(0)
        self._ck(self._lib.kdf_profile_stages(self._h, ms, byref(n)))
        return list(ms), n.value

    _PATHS = ("direct", "binned", "-", "sieve")
    _STAGES = ["kb_slabsort_kernel", "kb_groupsum+kb_plan", "kb_piecesort_pipe_kernel", "kb_bucket_kernel"]

    def last_count_path(self) -> str:
        """Which pipeline the last count call took: direct / binned / sieve."""
        return self._PATHS[self.get_stat("last_count_path")]

    def profile_stage_names(self):
        return list(self._STAGES)

    # -- count / filter ----------------------------------------------------
    def count(self, stream: ReadStream):
        self._ck(self._lib.kdf_count_reads(self._h, _vp(stream.packed), _vp(stream.invalid), stream.n_bases))
        return self

    def upload_async(self, slot: int, stream: ReadStream):
        """Copy a host batch into device staging slot 0 / 1 on the engine's copy stream (asynchronous when the
        arrays are pinned, reads.PinnedBatches); count it later with count_uploaded(slot)."""
        self._ck(self._lib.kdf_upload_reads_async(self._h, int(slot), _vp(stream.packed), _vp(stream.invalid), stream.n_bases))
        return self

    def count_uploaded(self, slot: int, filtered: bool = False):
        self._ck(self._lib.kdf_count_uploaded(self._h, int(slot), 1 if filtered else 0))
        return self

    def count_dev(self, d_packed: int, d_invalid: int, n_bases: int):
        """Stream resident in HBM (raw device pointers, padded per stream_words).
        The engine launches on its own stream unless set_stream() was called:
        the buffers must be complete (synchronise the producer) before the call."""
        self._ck(self._lib.kdf_count_reads_dev(self._h, c_void_p(d_packed), c_void_p(d_invalid), int(n_bases)))
        return self

    def add_pairs(self, lo: np.ndarray, hi: Optional[np.ndarray] = None, counts: Optional[np.ndarray] = None):
        """Insert-or-add (key, count) pairs (index load / `jellyfish merge`)."""
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = np.ascontiguousarray(hi, dtype=np.uint64) if self.wide else None
        cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
        self._ck(self._lib.kdf_add_pairs(self._h, _vp(lo), _vp(hi), _vp(cnt), len(lo)))
        return self

    def add_pairs_dev(self, d_lo: int, d_hi: Optional[int], d_counts: Optional[int], n: int):
        self._ck(self._lib.kdf_add_pairs_dev(self._h, c_void_p(d_lo), c_void_p(d_hi) if d_hi else None,
                                             c_void_p(d_counts) if d_counts else None, int(n)))
        return self

    def set_counts_dev(self, d_lo: int, d_hi: Optional[int], d_counts: int, n: int):
        """The count of every listed (stored) key becomes d_counts[i] (``kdf_set_counts_dev``)."""
        self._ck(self._lib.kdf_set_counts_dev(self._h, c_void_p(d_lo), c_void_p(d_hi) if d_hi else None, c_void_p(d_counts), int(n)))
        return self

    def add_pairs_multi_dev(self, segments):
        """Sum several device-resident segments of (lo ptr, hi ptr or None, counts ptr, n) into the table in ONE call
        (the owner's half of the multi-GPU merge, ``kdf_add_pairs_multi_dev``)."""
        segs = [s for s in segments if s[3]]
        if not segs:
            return self
        m = len(segs)
        lo = (c_void_p * m)(*[s[0] for s in segs])
        hi = (c_void_p * m)(*[s[1] or None for s in segs])
        cnt = (c_void_p * m)(*[s[2] for s in segs])
        n = (c_uint64 * m)(*[int(s[3]) for s in segs])
        self._ck(self._lib.kdf_add_pairs_multi_dev(self._h, m, lo, hi if self.wide else None, cnt, n))
        return self

    def load_filter(self, lo: np.ndarray, hi: Optional[np.ndarray] = None):
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        if self.wide:
            if hi is None:
                raise ValueError("wide keys (k > 32) need the hi words")
            hi = np.ascontiguousarray(hi, dtype=np.uint64)
        else:
            hi = None
        self._ck(self._lib.kdf_load_filter(self._h, _vp(lo), _vp(hi), len(lo)))
        return self

    def load_filter_dev(self, d_lo: int, d_hi: Optional[int], n: int):
        """Filter keys already resident in HBM (raw device pointers)."""
        self._ck(self._lib.kdf_load_filter_dev(self._h, c_void_p(d_lo), c_void_p(d_hi) if d_hi else None, int(n)))
        return self

    def reset_counts(self):
        """Zero every count, keep the keys (the same filter against the next parent)."""
        self._ck(self._lib.kdf_reset_counts(self._h))
        return self

    def count_filtered(self, stream: ReadStream):
        self._ck(self._lib.kdf_count_reads_filtered(self._h, _vp(stream.packed), _vp(stream.invalid),
                                                    stream.n_bases))
        return self

    def count_filtered_dev(self, d_packed: int, d_invalid: int, n_bases: int):
        self._ck(self._lib.kdf_count_reads_filtered_dev(self._h, c_void_p(d_packed), c_void_p(d_invalid),
                                                        int(n_bases)))
        return self

    # -- query / dump ------------------------------------------------------
    def query(self, lo: np.ndarray, hi: Optional[np.ndarray] = None) -> np.ndarray:
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = np.ascontiguousarray(hi, dtype=np.uint64) if (self.wide and hi is not None) else None
        if self.wide and hi is None:
            raise ValueError("wide keys (k > 32) need the hi words")
        out = np.zeros(len(lo), dtype=np.uint32)
        self._ck(self._lib.kdf_query(self._h, _vp(lo), _vp(hi), len(lo), _vp(out)))
        return out

    def query_dev(self, d_lo: int, d_hi: Optional[int], n: int, d_out: int):
        self._ck(self._lib.kdf_query_dev(self._h, c_void_p(d_lo), c_void_p(d_hi) if d_hi else None, int(n),
                                         c_void_p(d_out)))

    def count_ge(self, min_count: int) -> int:
        n = c_uint64(0)
        self._ck(self._lib.kdf_count_ge(self._h, int(min_count), byref(n)))
        return n.value

    def export_ge(self, min_count: int = 0):
        """(lo, hi, counts) of entries with count >= min_count, ascending key order."""
        n = self.count_ge(min_count)
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        cnt = np.zeros(n, np.uint32)
        got = c_uint64(0)
        self._ck(self._lib.kdf_export_ge(self._h, int(min_count), _vp(lo), _vp(hi), _vp(cnt), n, byref(got)))
        if got.value != n:
            raise _native.KdfError(_native.KDF_ERR_STATE, "export size changed between passes")
        return lo, hi, cnt

    def export_ge_dev(self, min_count: int, d_lo: int, d_hi: Optional[int], d_cnt: Optional[int], cap: int,
                      sorted_: bool = False) -> int:
        """Dump into caller-owned device buffers; returns the number of entries."""
        n = c_uint64(0)
        self._ck(self._lib.kdf_export_ge_dev(self._h, int(min_count), c_void_p(d_lo),
                                             c_void_p(d_hi) if d_hi else None,
                                             c_void_p(d_cnt) if d_cnt else None, int(cap),
                                             1 if sorted_ else 0, byref(n)))
        return n.value

    def export_parts_dev(self, min_count: int, parts: int, d_lo: int, d_hi: Optional[int], d_cnt: Optional[int],
                         cap: int):
        """Dump grouped by owner rank (distributed.owner_of) into caller-owned device
        buffers; returns (entries, per-owner counts)."""
        n = c_uint64(0)
        counts = (c_uint64 * int(parts))()
        self._ck(self._lib.kdf_export_parts_dev(self._h, int(min_count), int(parts), c_void_p(d_lo),
                                                c_void_p(d_hi) if d_hi else None,
                                                c_void_p(d_cnt) if d_cnt else None, int(cap), counts, byref(n)))
        return n.value, [int(x) for x in counts]

    def export_parts_packed_dev(self, min_count: int, parts: int, d_buf: int, cap_bytes: int):
        """The owner-ordered dump written into the packed all-to-all layout; returns (entries, per-owner counts,
        segment byte offsets [parts + 1])."""
        n = c_uint64(0)
        counts = (c_uint64 * int(parts))()
        offs = (c_uint64 * (int(parts) + 1))()
        self._ck(self._lib.kdf_export_parts_packed_dev(self._h, int(min_count), int(parts), c_void_p(d_buf), int(cap_bytes),
                                                       counts, offs, byref(n)))
        return n.value, [int(x) for x in counts], [int(x) for x in offs]

    # -- Module-3 scan -----------------------------------------------------
    def scan(self, stream: ReadStream, want_distinct: bool = True):
        """-> (hit_bits uint64[mask words], distinct uint32[n_reads] or None)."""
        _, mw = stream_words(stream.n_bases)
        hits = np.zeros(mw, np.uint64)
        distinct = np.zeros(stream.n_reads, np.uint32) if want_distinct else None
        offs = np.ascontiguousarray(stream.offsets, dtype=np.int64) if want_distinct else None
        self._ck(self._lib.kdf_scan_reads(self._h, _vp(stream.packed), _vp(stream.invalid), stream.n_bases,
                                          _vp(offs), stream.n_reads if want_distinct else 0,
                                          _vp(hits), _vp(distinct)))
        return hits, distinct

    def scan_dev(self, d_packed: int, d_invalid: int, n_bases: int, d_hits: int):
        self._ck(self._lib.kdf_scan_reads_dev(self._h, c_void_p(d_packed), c_void_p(d_invalid), int(n_bases),
                                              c_void_p(d_hits)))


def hit_positions(hit_bits: np.ndarray, start: int, end: int) -> np.ndarray:
    """Window-start offsets (relative to ``start``) whose hit bit is set in [start, end)."""
    w0, w1 = start >> 6, (end + 63) >> 6
    bits = np.unpackbits(hit_bits[w0:w1].view(np.uint8), bitorder="little")
    lo = start - (w0 << 6)
    return np.nonzero(bits[lo:lo + (end - start)])[0]
