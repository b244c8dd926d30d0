"""CPU ORACLE -- test infrastructure, not the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The shipped package ``kmer_denovo_filter_amd``
never does; its product path fails loudly when the HIP library is missing.

Parity status: PINNED (see ``tests/test_oracle_golden.py``): the restatement
below reproduces the real-Jellyfish fixture ``mini_ref.fa.k31.jf`` bit for bit,
the discovery chain 51125 -> 6679 -> 630 and the 195 / 11 informative-read
counts of ``tests/example_output_discovery/giab_discovery.metrics.json``.

Three layers, each citing the reference (paths relative to /root/reference):

* pure-Python string restatement (tiny inputs; mirrors the reference line by
  line in *behaviour*): ``canonicalize`` / ``reverse_complement`` /
  ``extract_read_kmers``  <- src/kmer_denovo_filter/kmer_utils.py:15-38,91-121
* ``OracleTable``: ctypes binding of ``kdf_oracle.c`` (hash-map restatement of
  jellyfish count -C [--if] / dump -c -L / query / Module-3 scan)
* independent file readers used to drive the goldens: BGZF/BAM via ``gzip`` +
  ``struct`` with ``samtools fasta -F 0xD00`` semantics, FASTA, and the Jellyfish
  ``binary/sorted`` reader.
"""
from __future__ import annotations

import ctypes
import gzip
import json
import os
import struct
import subprocess
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkdf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile ``kdf_oracle.c`` -> ``libkdf_oracle.so`` (gcc, seconds)."""
    src = os.path.join(_HERE, "kdf_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.run(
        ["gcc", "-O2", "-fPIC", "-std=gnu11", "-shared", "-o", _LIB_PATH, src,
         "-lpthread"],
        check=True,
    )
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    vp, i64, u64, u32, ci = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                             ctypes.c_uint32, ctypes.c_int)
    lib.kdfo_create.restype = vp
    lib.kdfo_create.argtypes = [ci, u64]
    lib.kdfo_destroy.argtypes = [vp]
    lib.kdfo_size.restype = u64
    lib.kdfo_size.argtypes = [vp]
    lib.kdfo_canonical_ascii.argtypes = [ctypes.c_char_p, ci, vp, vp]
    for name in ("kdfo_count_reads", "kdfo_count_reads_filtered"):
        getattr(lib, name).argtypes = [vp, vp, vp, i64]
    lib.kdfo_count_reads_mt.argtypes = [vp, vp, vp, i64, ci]
    lib.kdfo_load_filter.argtypes = [vp, vp, vp, i64]
    lib.kdfo_query.argtypes = [vp, vp, vp, i64, vp]
    lib.kdfo_export_ge.restype = i64
    lib.kdfo_export_ge.argtypes = [vp, u32, vp, vp, vp]
    lib.kdfo_scan_reads.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.kdfo_count_tally_mt.argtypes = [vp, vp, i64, ci, ci, u32, vp]
    lib.kdfo_count_windows.restype = i64
    lib.kdfo_count_windows.argtypes = [vp, vp, i64, ci]
    _lib = lib
    return lib


# --------------------------------------------------------------------------
# pure-Python string restatement (kmer_utils.py:15-38, 91-121)
# --------------------------------------------------------------------------

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")      # kmer_utils.py:15


def reverse_complement(seq: str) -> str:            # kmer_utils.py:30-32
    return seq.translate(_COMP)[::-1]


def canonicalize(kmer: str) -> str:                 # kmer_utils.py:35-38
    rc = kmer.translate(_COMP)[::-1]
    return kmer if kmer < rc else rc


def extract_read_kmers(seq: str, k: int):           # kmer_utils.py:91-121
    """(canon_at_pos, unique_candidates) with first-seen order."""
    if len(seq) < k:
        return {}, []
    s = seq.upper()
    canon_at_pos, cands = {}, []
    for i in range(len(s) - k + 1):
        w = s[i:i + k]
        if "N" in w:
            continue
        c = canonicalize(w)
        canon_at_pos[i] = c
        cands.append(c)
    return canon_at_pos, list(dict.fromkeys(cands))


_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_BASES = "ACGT"


def kmer_to_int(kmer: str) -> int:
    """Jellyfish key encoding: A=0 C=1 G=2 T=3, leftmost base most significant."""
    v = 0
    for ch in kmer.upper():
        v = (v << 2) | _CODE[ch]
    return v


def int_to_kmer(v: int, k: int) -> str:
    return "".join(_BASES[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def py_count(reads: Iterable[str], k: int, filt: Optional[set] = None) -> Dict[str, int]:
    """Dict counter with Jellyfish's window rule (any non-ACGT byte breaks the
    window; case-insensitive).  ``filt`` restates ``--if``."""
    out: Dict[str, int] = {}
    if filt is not None:
        out = {canonicalize(f.upper()): 0 for f in filt}
    for r in reads:
        s = r.upper()
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if any(ch not in _CODE for ch in w):
                continue
            c = canonicalize(w)
            if filt is None:
                out[c] = out.get(c, 0) + 1
            elif c in out:
                out[c] += 1
    return out


# --------------------------------------------------------------------------
# ctypes table
# --------------------------------------------------------------------------

def concat_reads(reads: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """ASCII reads -> (uint8 buffer, int64 offsets[n+1])."""
    bs = [r if isinstance(r, (bytes, bytearray)) else r.encode() for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs])
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return buf, offs


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleTable:
    """Exact (key -> uint32 count) table; keys are 2k-bit ints (k <= 64)."""

    def __init__(self, k: int, cap_hint: int = 1024):
        self.lib = _load()
        self.k = int(k)
        self.wide = self.k > 32
        self.h = self.lib.kdfo_create(self.k, int(cap_hint))
        if not self.h:
            raise ValueError(f"oracle: unsupported k={k}")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.kdfo_destroy(self.h)
            self.h = None

    def __len__(self):
        return int(self.lib.kdfo_size(self.h))

    @staticmethod
    def _buf(reads):
        if isinstance(reads, tuple):
            return reads
        return concat_reads(reads)

    def count_reads(self, reads, threads: int = 1):
        buf, offs = self._buf(reads)
        if threads > 1:
            self.lib.kdfo_count_reads_mt(self.h, _p(buf), _p(offs), len(offs) - 1, threads)
        else:
            self.lib.kdfo_count_reads(self.h, _p(buf), _p(offs), len(offs) - 1)
        return self

    def load_filter(self, lo: np.ndarray, hi: Optional[np.ndarray] = None):
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = None if hi is None else np.ascontiguousarray(hi, dtype=np.uint64)
        self.lib.kdfo_load_filter(self.h, _p(lo), _p(hi), len(lo))
        return self

    def count_reads_filtered(self, reads):
        buf, offs = self._buf(reads)
        self.lib.kdfo_count_reads_filtered(self.h, _p(buf), _p(offs), len(offs) - 1)
        return self

    def query(self, lo: np.ndarray, hi: Optional[np.ndarray] = None) -> np.ndarray:
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = None if hi is None else np.ascontiguousarray(hi, dtype=np.uint64)
        out = np.zeros(len(lo), dtype=np.uint32)
        self.lib.kdfo_query(self.h, _p(lo), _p(hi), len(lo), _p(out))
        return out

    def export_ge(self, min_count: int = 0):
        """Sorted (lo, hi, counts) of entries with count >= min_count."""
        n = self.lib.kdfo_export_ge(self.h, min_count, None, None, None)
        lo = np.zeros(n, dtype=np.uint64)
        hi = np.zeros(n, dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        self.lib.kdfo_export_ge(self.h, min_count, _p(lo), _p(hi), _p(cnt))
        return lo, hi, cnt

    def scan_reads(self, reads):
        """-> (hit bitmap over concatenated positions as bool array, distinct per read)."""
        buf, offs = self._buf(reads)
        nbits = int(offs[-1])
        bits = np.zeros((nbits + 7) // 8 + 1, dtype=np.uint8)
        distinct = np.zeros(len(offs) - 1, dtype=np.uint32)
        self.lib.kdfo_scan_reads(self.h, _p(buf), _p(offs), len(offs) - 1, _p(bits), _p(distinct))
        hit = np.unpackbits(bits, bitorder="little")[:nbits].astype(bool)
        return hit, distinct


def count_windows(reads, k: int) -> int:
    buf, offs = reads if isinstance(reads, tuple) else concat_reads(reads)
    return int(_load().kdfo_count_windows(_p(buf), _p(offs), len(offs) - 1, k))


def count_tally_mt(reads, k: int, threads: int, min_count: int = 3):
    """(distinct, sum of counts, k-mers with count >= min_count) of a full count of ``reads``:
    the partitioned multi-core count that bench.py times as its CPU baseline."""
    buf, offs = reads if isinstance(reads, tuple) else concat_reads(reads)
    out = np.zeros(3, dtype=np.uint64)
    if _load().kdfo_count_tally_mt(_p(buf), _p(offs), len(offs) - 1, k, threads, min_count, _p(out)) != 0:
        raise MemoryError("oracle: kdfo_count_tally_mt failed")
    return int(out[0]), int(out[1]), int(out[2])


def canonical_key(kmer: str) -> int:
    lo, hi = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = _load().kdfo_canonical_ascii(kmer.encode(), len(kmer), ctypes.byref(lo), ctypes.byref(hi))
    if rc != 0:
        raise ValueError("non-ACGT base in k-mer")
    return (hi.value << 64) | lo.value


# --------------------------------------------------------------------------
# file readers that drive the goldens
# --------------------------------------------------------------------------

def read_fasta(path: str) -> List[Tuple[str, str]]:
    out, name, chunks = [], None, []
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(chunks)))
                name, chunks = line[1:].split()[0] if len(line) > 1 else "", []
            elif line:
                chunks.append(line)
    if name is not None:
        out.append((name, "".join(chunks)))
    return out


def read_kmer_fasta(path: str) -> List[str]:
    """``>i\\nKMER\\n`` intermediate files (utils.py:150-170)."""
    out = []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line and not line.startswith(">"):
                out.append(line)
    return out


def read_jf_binary_sorted(path: str):
    """Jellyfish ``binary/sorted`` -> (header dict, keys object-array of ints, counts).

    Layout: 9 ASCII digits = length of the JSON header
    including NUL padding; records of ceil(key_len/8) key bytes LE +
    counter_len count bytes LE, in hash order.
    """
    with open(path, "rb") as fh:
        raw = fh.read()
    hlen = int(raw[:9])
    header = json.loads(raw[9:9 + hlen].rstrip(b"\0").decode())
    if header.get("format") != "binary/sorted":
        raise ValueError(f"unsupported Jellyfish format {header.get('format')!r}")
    kb = (header["key_len"] + 7) // 8
    cb = header["counter_len"]
    data = raw[9 + hlen:]
    rec = kb + cb
    n = len(data) // rec
    keys, counts = [], np.zeros(n, dtype=np.uint64)
    for i in range(n):
        r = data[i * rec:(i + 1) * rec]
        keys.append(int.from_bytes(r[:kb], "little"))
        counts[i] = int.from_bytes(r[kb:], "little")
    return header, keys, counts


_SEQ_NT16 = "=ACMGRSVTWYHKDBN"


class BamRecord:
    __slots__ = ("qname", "flag", "ref_id", "pos", "mapq", "cigar", "seq",
                 "has_qual", "next_ref_id", "next_pos", "tlen", "tags")

    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def is_secondary(self): return bool(self.flag & 0x100)
    @property
    def is_duplicate(self): return bool(self.flag & 0x400)
    @property
    def is_supplementary(self): return bool(self.flag & 0x800)


def read_bam(path: str) -> Tuple[List[Tuple[str, int]], Iterator[BamRecord]]:
    """Minimal BAM reader: BGZF is a series of gzip members, which ``gzip``
    concatenates transparently.  Returns (references, record iterator)."""
    fh = gzip.open(path, "rb")
    if fh.read(4) != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack("<i", fh.read(4))
    fh.read(l_text)
    n_ref, = struct.unpack("<i", fh.read(4))
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack("<i", fh.read(4))
        name = fh.read(l_name)[:-1].decode()
        l_ref, = struct.unpack("<i", fh.read(4))
        refs.append((name, l_ref))

    def it():
        while True:
            hdr = fh.read(4)
            if len(hdr) < 4:
                break
            bs, = struct.unpack("<i", hdr)
            blk = fh.read(bs)
            (ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, nref, npos,
             tlen) = struct.unpack("<iiBBHHHiiii", blk[:32])
            r = BamRecord()
            o = 32
            r.qname = blk[o:o + l_rn - 1].decode(); o += l_rn
            r.cigar = [(c & 0xF, c >> 4) for c in struct.unpack(f"<{n_cig}I", blk[o:o + 4 * n_cig])]
            o += 4 * n_cig
            sb = blk[o:o + (l_seq + 1) // 2]; o += (l_seq + 1) // 2
            chars = []
            for b in sb:
                chars.append(_SEQ_NT16[b >> 4]); chars.append(_SEQ_NT16[b & 0xF])
            r.seq = "".join(chars[:l_seq])
            r.has_qual = l_seq > 0 and blk[o] != 0xFF
            o += l_seq
            r.tags = blk[o:]
            r.flag, r.ref_id, r.pos, r.mapq = flag, ref_id, pos, mapq
            r.next_ref_id, r.next_pos, r.tlen = nref, npos, tlen
            yield r
        fh.close()

    return refs, it()


def samtools_fasta_reads(path: str, flag_off: int = 0xD00) -> List[str]:
    """Bases emitted by ``samtools fasta -F 0xD00`` (strand ignored: irrelevant
    under -C).  Records with any ``flag_off`` bit are dropped; each run of
    consecutive same-QNAME records is collapsed to at most one record per read
    part (READ1 / READ2 / other), a record with qualities beating one without,
    first one winning ties (reference call sites
    core/jellyfish_wrappers.py:159-165, discovery/pipeline.py:106-112)."""
    _, recs = read_bam(path)
    out: List[str] = []
    cur, best, score = None, [None, None, None], [-1, -1, -1]

    def flush():
        for s in best:
            if s is not None:
                out.append(s)

    for r in recs:
        if r.flag & flag_off:
            continue
        if cur is None or r.qname != cur:
            flush()
            cur, best, score = r.qname, [None, None, None], [-1, -1, -1]
        r1, r2 = bool(r.flag & 0x40), bool(r.flag & 0x80)
        part = 1 if (r1 and not r2) else 2 if (r2 and not r1) else 0
        sc = 2 if r.has_qual else 1
        if sc > score[part]:
            best[part], score[part] = r.seq, sc
    flush()
    return out


# --------------------------------------------------------------------------
# stage restatements used by the golden tests
# --------------------------------------------------------------------------

def keys_to_arrays(keys: Sequence[int]):
    lo = np.array([k & 0xFFFFFFFFFFFFFFFF for k in keys], dtype=np.uint64)
    hi = np.array([k >> 64 for k in keys], dtype=np.uint64)
    return lo, hi


def discovery_chain(child_reads, mother_reads, father_reads, ref_table: "OracleTable",
                    k: int, min_child_count: int = 3, parent_max_count: int = 0):
    """discovery/pipeline.py:69-612 -- Module 1 + ref subtraction + Module 2.

    Returns dict of sorted key arrays per stage: candidates (count >=
    min_child_count), non_ref (ref count == 0), after_mother, proband_unique
    (parent count <= parent_max_count, mother first then father on survivors).
    """
    child = OracleTable(k, 1 << 16).count_reads(child_reads)
    lo, hi, _ = child.export_ge(min_child_count)                 # dump -c -L n
    refc = ref_table.query(lo, hi)                               # query ref.jf
    keep = refc == 0                                             # == "0"
    nlo, nhi = lo[keep], hi[keep]
    stages = {"candidates": (lo, hi), "non_ref": (nlo, nhi)}
    cur = (nlo, nhi)
    for label, reads in (("after_mother", mother_reads), ("proband_unique", father_reads)):
        if len(cur[0]) == 0:
            stages[label] = cur
            continue
        t = OracleTable(k, len(cur[0]) * 2).load_filter(*cur).count_reads_filtered(reads)
        c = t.query(*cur)
        keep = c <= parent_max_count                             # int(cnt) <= parent_max_count
        cur = (cur[0][keep], cur[1][keep])
        stages[label] = cur
    return stages


def module3_scan(child_bam: str, proband_lo, proband_hi, k: int, min_dk_per_read: int):
    """core/bam_scanner.py:340-474 + discovery/pipeline.py:733-860 restated.

    One task per contig plus one for unplaced unmapped reads; SECONDARY and
    DUPLICATE skipped, supplementary kept, no QNAME collapse; a read is
    informative when its number of distinct hit k-mers >= min_dk_per_read;
    informative records are deduplicated by (qname, is_supplementary) -- first
    per task, then across tasks.  Returns (total_informative,
    unmapped_informative, per-record hit info list).
    """
    refs, recs = read_bam(child_bam)
    # jellyfish_wrappers.py:369-436: the proband index is `jellyfish count` of the
    # k-mer FASTA, so every proband-unique k-mer is present with count >= 1.
    kmers = [int_to_kmer((int(h) << 64) | int(l), k) for l, h in zip(proband_lo, proband_hi)]
    table = OracleTable(k, max(1024, 2 * len(kmers))).count_reads(kmers)
    tasks: Dict[int, List[BamRecord]] = {}
    for r in recs:
        if r.is_secondary or r.is_duplicate:
            continue
        tasks.setdefault(r.ref_id, []).append(r)
    order = [i for i in range(len(refs)) if i in tasks] + ([-1] if -1 in tasks else [])
    seen_global, total_mapped, unmapped_inf, hits_out = set(), 0, 0, []
    for ref_id in order:
        rl = tasks[ref_id]
        buf, offs = concat_reads([r.seq for r in rl])
        hit, distinct = table.scan_reads((buf, offs))
        seen_local = set()
        for i, r in enumerate(rl):
            if distinct[i] < min_dk_per_read:
                continue
            key = (r.qname, r.is_supplementary)
            if key in seen_local:
                continue
            seen_local.add(key)
            if r.is_unmapped:
                unmapped_inf += 1
                continue
            if key in seen_global:
                continue
            total_mapped += 1
            hits_out.append((r, np.nonzero(hit[offs[i]:offs[i + 1]])[0], int(distinct[i])))
        seen_global |= seen_local
    return total_mapped + unmapped_inf, unmapped_inf, hits_out
