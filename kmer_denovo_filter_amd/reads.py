"""Read streams: the 2-bit-packed base stream + invalid mask the engine consumes,
and the host feeders that produce them (ASCII reads, BAM, FASTA).

Replaces the ``samtools fasta -F 0xD00 | ...`` half of the reference's pipes
(core/jellyfish_wrappers.py:159-165, discovery/pipeline.py:106-112,369-375)
and pysam's read iteration in Module 3 (core/bam_scanner.py:405-414).
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_char_p, c_int32, c_int64, c_uint16, c_uint64, c_void_p, POINTER
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Sequence

import numpy as np

from . import _native

# samtools fasta -F 0xD00: SECONDARY | DUPLICATE | SUPPLEMENTARY
FLAG_OFF_SAMTOOLS_FASTA = 0xD00
# Module 3 (bam_scanner.py:406-409): SECONDARY | DUPLICATE only
FLAG_OFF_MODULE3 = 0x500


def stream_words(n_bases: int):
    pw, mw = c_uint64(0), c_uint64(0)
    _native.load().kdf_stream_words(int(n_bases), byref(pw), byref(mw))
    return pw.value, mw.value


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_void_p)


@dataclass
class ReadStream:
    """One batch of reads as a packed stream (layout: include/kdf.h)."""
    packed: np.ndarray            # uint64, stream_words()[0] words (zero padded)
    invalid: np.ndarray           # uint64, stream_words()[1] words (tail = all ones)
    n_bases: int                  # stream positions, separators included
    offsets: np.ndarray           # int64[n_reads+1]: start of each read in the stream
    # per-read metadata (BAM feeders only)
    flags: Optional[np.ndarray] = None
    ref_ids: Optional[np.ndarray] = None
    positions: Optional[np.ndarray] = None
    name_buf: Optional[bytes] = None          # NUL-terminated names, back to back
    name_offsets: Optional[np.ndarray] = None
    ordinals: Optional[np.ndarray] = None     # uint64[n_reads]: record number in the BAM file (write_bam_subset)
    # alignment details (bam_reader(..., want_aux=True))
    cigar: Optional[np.ndarray] = None        # uint32, BAM encoding (len << 4 | op), all records back to back
    cigar_offsets: Optional[np.ndarray] = None  # int64[n_reads + 1]
    sa_buf: Optional[bytes] = None
    sa_offsets: Optional[np.ndarray] = None   # int64[n_reads], -1 = no SA tag
    quals: Optional[np.ndarray] = None        # uint8 base qualities, all records back to back
    qual_offsets: Optional[np.ndarray] = None  # int64[n_reads + 1]
    mapq: Optional[np.ndarray] = None         # uint8[n_reads]

    def qualities(self, i: int):
        """uint8 base qualities of record i, or None when the BAM stores none (0xFF)."""
        q = self.quals[int(self.qual_offsets[i]):int(self.qual_offsets[i + 1])]
        return None if (len(q) and q[0] == 0xFF) else q

    def cigartuples(self, i: int):
        """[(op, length), ...] of record i, as pysam's ``cigartuples``."""
        c = self.cigar[int(self.cigar_offsets[i]):int(self.cigar_offsets[i + 1])]
        return [(int(x) & 0xF, int(x) >> 4) for x in c]

    def sa_tag(self, i: int):
        o = int(self.sa_offsets[i])
        if o < 0:
            return None
        return self.sa_buf[o:self.sa_buf.index(b"\0", o)].decode()

    def name(self, i: int) -> str:
        o = int(self.name_offsets[i])
        return self.name_buf[o:self.name_buf.index(b"\0", o)].decode()

    @property
    def names(self) -> Optional[List[str]]:
        """All read names (decoded on demand: Module 3 only needs the few informative ones)."""
        if self.name_buf is None:
            return None
        return [x.decode() for x in self.name_buf.split(b"\0")[:self.n_reads]]

    @property
    def n_reads(self) -> int:
        return len(self.offsets) - 1

    def read_lengths(self) -> np.ndarray:
        """Bases per read (the separator after each read is not counted)."""
        return np.diff(self.offsets) - 1

    @staticmethod
    def empty() -> "ReadStream":
        pw, mw = stream_words(0)
        return ReadStream(np.zeros(pw, np.uint64), np.full(mw, ~np.uint64(0), np.uint64), 0,
                          np.zeros(1, np.int64))

    @staticmethod
    def from_ascii(buf: np.ndarray, offs: np.ndarray) -> "ReadStream":
        """ASCII records (uint8 buffer + int64 offsets[n+1]) -> stream."""
        lib = _native.load()
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        n_reads = len(offs) - 1
        total = int(offs[-1] - offs[0]) + n_reads if n_reads else 0
        pw, mw = stream_words(total)
        packed = np.zeros(pw, np.uint64)
        invalid = np.full(mw, ~np.uint64(0), np.uint64)
        so = np.zeros(n_reads + 1, np.int64)
        nb = c_uint64(0)
        rc = lib.kdf_pack_reads(_vp(buf), _vp(offs), n_reads, _vp(packed), _vp(invalid), _vp(so), byref(nb))
        _native.check(rc)
        # words the packer did not touch keep their padding defaults
        return ReadStream(packed, invalid, nb.value, so)

    @staticmethod
    def from_strings(reads: Sequence) -> "ReadStream":
        bs = [r if isinstance(r, (bytes, bytearray)) else r.encode() for r in reads]
        offs = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs])
        buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
        return ReadStream.from_ascii(buf, offs)


class _Reader:
    """Iterator over ReadStream batches from a native kdf_reader."""

    def __init__(self, handle, max_bases: int, max_reads: int, want_meta: bool, want_aux: bool = False,
                 is_bam: bool = False):
        self._h = handle
        self.max_bases = int(max_bases)
        self.max_reads = int(max_reads)
        self.want_meta = want_meta or want_aux
        self.want_aux = want_aux
        self.is_bam = is_bam
        self._lib = _native.load()
        if want_aux:
            _native.check_reader(self._lib.kdf_reader_want_aux(self._h, 1), self._h)

    def references(self) -> List[str]:
        n = self._lib.kdf_reader_ref_count(self._h)
        return [self._lib.kdf_reader_ref_name(self._h, i).decode() for i in range(max(n, 0))]

    def close(self):
        if self._h:
            self._lib.kdf_reader_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _next(self, packed, invalid, so):
        """One batch into the given arrays -> ReadStream (views of them), or None at end of file."""
        n_reads, nb = c_int64(0), c_uint64(0)
        rc = self._lib.kdf_reader_next(self._h, self.max_bases, self.max_reads, _vp(packed), _vp(invalid),
                                       _vp(so), byref(n_reads), byref(nb))
        _native.check_reader(rc, self._h)
        n = n_reads.value
        if n == 0:
            return None
        pw_used, mw_used = stream_words(nb.value)
        return ReadStream(packed[:pw_used], invalid[:mw_used], nb.value, so[:n + 1].copy())

    def next_into(self, packed: np.ndarray, invalid: np.ndarray):
        """Sequence-only batch into caller-owned arrays (e.g. pinned ones, PinnedBatches) of stream_words(max_bases)
        words; no per-record metadata.  None at end of file."""
        if not self._h:
            return None
        pw, mw = stream_words(self.max_bases)
        if len(packed) < pw or len(invalid) < mw:
            raise ValueError("batch arrays are smaller than stream_words(max_bases)")
        so = np.empty(self.max_reads + 1, np.int64)
        st = self._next(packed, invalid, so)
        if st is None:
            self.close()
        return st

    def __iter__(self) -> Iterator[ReadStream]:
        lib = self._lib
        pw, mw = stream_words(self.max_bases)
        while self._h:
            packed = np.empty(pw, np.uint64)           # the native reader initialises both arrays
            invalid = np.empty(mw, np.uint64)
            so = np.empty(self.max_reads + 1, np.int64)
            st = self._next(packed, invalid, so)
            if st is None:
                break
            n = st.n_reads
            if self.want_meta:
                f, r, p = POINTER(c_uint16)(), POINTER(c_int32)(), POINTER(c_int32)()
                nbuf, noff = c_char_p(), POINTER(c_int64)()
                lib.kdf_reader_last_meta(self._h, byref(f), byref(r), byref(p), byref(nbuf), byref(noff))
                st.flags = np.ctypeslib.as_array(f, (n,)).copy()
                st.ref_ids = np.ctypeslib.as_array(r, (n,)).copy()
                st.positions = np.ctypeslib.as_array(p, (n,)).copy()
                offs = np.ctypeslib.as_array(noff, (n,)).copy()
                base = ctypes.cast(nbuf, c_void_p).value
                last = int(offs[-1])
                end = last + len(ctypes.string_at(base + last)) + 1
                st.name_buf = ctypes.string_at(base, end)
                st.name_offsets = offs
                if self.is_bam:
                    od = POINTER(c_uint64)()
                    _native.check_reader(lib.kdf_reader_last_ordinals(self._h, byref(od)), self._h)
                    st.ordinals = np.ctypeslib.as_array(od, (n,)).copy()
            if self.want_aux:
                from ctypes import c_uint32
                cg, cgo = POINTER(c_uint32)(), POINTER(c_int64)()
                sab, sao = c_char_p(), POINTER(c_int64)()
                _native.check_reader(lib.kdf_reader_last_aux(self._h, byref(cg), byref(cgo), byref(sab), byref(sao)),
                                     self._h)
                st.cigar_offsets = np.ctypeslib.as_array(cgo, (n + 1,)).copy()
                ncg = int(st.cigar_offsets[-1])
                st.cigar = np.ctypeslib.as_array(cg, (ncg,)).copy() if ncg else np.zeros(0, np.uint32)
                st.sa_offsets = np.ctypeslib.as_array(sao, (n,)).copy()
                valid = st.sa_offsets[st.sa_offsets >= 0]
                if len(valid):
                    sbase = ctypes.cast(sab, c_void_p).value
                    last = int(valid.max())
                    st.sa_buf = ctypes.string_at(sbase, last + len(ctypes.string_at(sbase + last)) + 1)
                else:
                    st.sa_buf = b""
                from ctypes import c_uint8
                qb, qo, mq = POINTER(c_uint8)(), POINTER(c_int64)(), POINTER(c_uint8)()
                _native.check_reader(lib.kdf_reader_last_quals(self._h, byref(qb), byref(qo), byref(mq)), self._h)
                st.qual_offsets = np.ctypeslib.as_array(qo, (n + 1,)).copy()
                nq = int(st.qual_offsets[-1])
                st.quals = np.ctypeslib.as_array(qb, (nq,)).copy() if nq else np.zeros(0, np.uint8)
                st.mapq = np.ctypeslib.as_array(mq, (n,)).copy()
            yield st
        self.close()


class PinnedBatches:
    """A ring of page-locked host batches (packed + invalid words for ``max_bases`` stream positions each): the H2D copy
    of one is asynchronous, so it can run while the GPU counts the batch before it and the reader fills the next."""

    def __init__(self, n: int, max_bases: int):
        self._lib = _native.load()
        self._ptrs = []
        self.batches = []
        pw, mw = stream_words(max_bases)
        try:
            for _ in range(n):
                arrs = []
                for words in (pw, mw):
                    p = c_void_p()
                    _native.check(self._lib.kdf_host_alloc(words * 8, byref(p)), None)
                    self._ptrs.append(p)
                    arrs.append(np.ctypeslib.as_array(ctypes.cast(p, POINTER(c_uint64)), (words,)))
                self.batches.append(tuple(arrs))
        except Exception:
            self.close()
            raise

    def close(self):
        self.batches = []
        for p in self._ptrs:
            self._lib.kdf_host_free(p)
        self._ptrs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


_pinned_cache = {}            # (ring, max_bases) -> PinnedBatches, kept for the life of the process: page-locking ~50 MB per
#                               batch costs more than counting it, and the child and both parents stream one after another


def _pinned_ring(ring: int, max_bases: int) -> PinnedBatches:
    key = (ring, max_bases)
    pb = _pinned_cache.get(key)
    if pb is None or not pb.batches:
        for old in list(_pinned_cache.values()):
            old.close()
        _pinned_cache.clear()
        pb = _pinned_cache[key] = PinnedBatches(ring, max_bases)
    return pb


def stream_batches_overlapped(engine, readers, filtered: bool, ring: int = 0) -> int:
    """Count every batch of ``readers`` (one reader, or several readers of disjoint ranges of one file:
    ``bam_reader(part=, parts=)``) on ``engine`` as a three-stage pipeline: one reader thread PER READER decodes batches
    into pinned buffers (the native reader releases the GIL; inflate, chunking and parsing of the ranges run side by
    side), the copy stream uploads batch i + 1, the engine counts batch i.  Batches arrive in any order: counting does
    not care.  Returns the number of reads."""
    import queue
    import threading
    if not isinstance(readers, (list, tuple)):
        readers = [readers]
    n_reads = 0
    ring = ring or len(readers) + 2
    pinned = _pinned_ring(ring, readers[0].max_bases)
    free, ready = queue.Queue(), queue.Queue()
    for i in range(ring):
        free.put(i)
    stop = threading.Event()

    def produce(reader):
        try:
            while not stop.is_set():
                i = free.get()
                if i is None:
                    free.put(None)                           # (pass the stop token on to the other producers)
                    break
                st = reader.next_into(*pinned.batches[i])
                if st is None:
                    free.put(i)                              # unused: back to the ring
                    break
                ready.put((i, st))
            ready.put(None)
        except BaseException as ex:  # noqa: BLE001 -- handed to the consumer
            ready.put(ex)

    threads = [threading.Thread(target=produce, args=(rd,), name="kdf-reader", daemon=True) for rd in readers]
    for th in threads:
        th.start()
    pending = None                                           # (slot, buffer) uploaded, not yet counted
    slot = 0
    live = len(threads)
    try:
        while live:
            item = ready.get()
            if item is None:
                live -= 1
                continue
            if isinstance(item, BaseException):
                raise item
            i, st = item
            engine.upload_async(slot, st)                    # returns at once: the buffer is pinned
            n_reads += st.n_reads
            if pending is not None:
                engine.count_uploaded(pending[0], filtered)
                free.put(pending[1])                         # count_uploaded waited (on the HOST) for this buffer's copy
            pending = (slot, i)
            slot ^= 1
        if pending is not None:
            engine.count_uploaded(pending[0], filtered)
            pending = None
    finally:
        stop.set()
        free.put(None)
        for th in threads:
            th.join()                                        # the native reader call always returns; never close a reader under it
        engine.synchronize()                                 # no copy may still read a pinned buffer the next caller will fill
    return n_reads


def bam_reader(path: str, flag_off: int = FLAG_OFF_SAMTOOLS_FASTA, collapse: bool = True,
               max_bases: int = 1 << 26, max_reads: int = 1 << 20, threads: int = 1,
               want_meta: bool = False, want_aux: bool = False, part: int = 0, parts: int = 1) -> _Reader:
    """``samtools fasta -F flag_off`` as an iterator of ReadStream batches.  ``part`` of ``parts``: one BGZF range of
    the file, cut on record (QNAME-run) boundaries -- the parts together are the whole file, each record once
    (``kdf_bam_open_range``: the shard of one rank, or of one reader pipeline inside a process)."""
    h = c_void_p()
    if parts > 1:
        rc = _native.load().kdf_bam_open_range(path.encode(), flag_off, 1 if collapse else 0, threads, int(part), int(parts), byref(h))
    else:
        rc = _native.load().kdf_bam_open(path.encode(), flag_off, 1 if collapse else 0, threads, byref(h))
    _native.check_reader(rc, None)
    return _Reader(h, max_bases, max_reads, want_meta, want_aux, is_bam=True)


def write_bam_subset(src_bam: str, dst_bam: str, ordinals, aux_fields=None, sort_and_index: bool = True,
                     threads: int = 1) -> int:
    """Copy the records ``ordinals`` (ReadStream.ordinals values) of ``src_bam`` into
    ``dst_bam``; ``aux_fields[i]`` (bytes, BAM-encoded optional fields) is appended to
    record i.  With ``sort_and_index``: coordinate sorted + ``.bai`` (the reference's
    ``pysam.sort`` + ``pysam.index``).  Returns the number of records written."""
    order = np.argsort(np.asarray(ordinals, dtype=np.uint64), kind="stable")
    od = np.ascontiguousarray(np.asarray(ordinals, dtype=np.uint64)[order])
    aux = offs = None
    if aux_fields is not None:
        parts = [bytes(aux_fields[i]) for i in order.tolist()]
        offs = np.zeros(len(parts) + 1, np.uint64)
        if parts:
            offs[1:] = np.cumsum([len(x) for x in parts])
        aux = np.frombuffer(b"".join(parts), np.uint8) if int(offs[-1]) else np.zeros(1, np.uint8)
    nw = c_uint64(0)
    rc = _native.load().kdf_bam_write_subset(src_bam.encode(), dst_bam.encode(), _vp(od), len(od), _vp(aux), _vp(offs),
                                             1 if sort_and_index else 0, threads, byref(nw))
    _native.check_reader(rc, None)
    return nw.value


def fasta_reader(path: str, k: int, max_bases: int = 1 << 26, max_reads: int = 1 << 16,
                 want_meta: bool = False) -> _Reader:
    h = c_void_p()
    rc = _native.load().kdf_fasta_open(path.encode(), int(k), byref(h))
    _native.check_reader(rc, None)
    return _Reader(h, max_bases, max_reads, want_meta)


# --------------------------------------------------------------------------
# k-mer string <-> key codec (Jellyfish encoding, A=0 C=1 G=2 T=3, MSB first)
# --------------------------------------------------------------------------

_ENC = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _ENC[_c] = _i
    _ENC[_c + 32] = _i          # lower case
_DEC = np.frombuffer(b"ACGT", dtype=np.uint8)


def kmers_to_keys(kmers: Sequence[str], k: int, canonical: bool = True):
    """K-mer strings -> (lo, hi) uint64 arrays of (canonical) keys."""
    n = len(kmers)
    lo = np.zeros(n, np.uint64)
    hi = np.zeros(n, np.uint64)
    if n == 0:
        return lo, hi
    raw = np.frombuffer("".join(kmers).encode(), dtype=np.uint8)
    if raw.size != n * k:
        raise ValueError("k-mer of wrong length in input")
    codes = _ENC[raw].reshape(n, k)
    if (codes > 3).any():
        raise ValueError("non-ACGT base in k-mer")
    codes = codes.astype(np.uint64)

    def pack(c):
        plo = np.zeros(n, np.uint64)
        phi = np.zeros(n, np.uint64)
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


def keys_to_kmers(lo: np.ndarray, hi: Optional[np.ndarray], k: int) -> List[str]:
    n = len(lo)
    if n == 0:
        return []
    lo = np.asarray(lo, np.uint64)
    hi = np.zeros(n, np.uint64) if hi is None else np.asarray(hi, np.uint64)
    out = np.empty((n, k), dtype=np.uint8)
    for i in range(k):
        sh = 2 * (k - 1 - i)
        src = (hi >> np.uint64(sh - 64)) if sh >= 64 else (lo >> np.uint64(sh))
        out[:, i] = _DEC[(src & np.uint64(3)).astype(np.intp)]
    flat = out.tobytes().decode()
    return [flat[i * k:(i + 1) * k] for i in range(n)]
