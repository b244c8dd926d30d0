"""Module 3 hot loop: mirror of the k-mer probing part of
``kmer_denovo_filter/core/bam_scanner.py`` (``_init_scan_worker`` :250-281,
``_scan_contig_for_hits`` JF branch :396-474).

The reference extracts every window of every read in Python, unions the k-mers
of 5000 reads and spawns one ``jellyfish query`` per batch.  Here a batch of
reads is one packed stream and ONE scan-kernel launch returns a hit bit per
window; reads are kept when their number of DISTINCT hit k-mers reaches
``min_distinct_kmers_per_read``.

What is replaced is the probe loop.  Mapping hits to reference coordinates and
SV metadata (``_process_informative_read`` :284-337, ``_collect_kmer_ref_positions``
:97-117) consumes the per-read hit positions returned here; that host
post-processing is the next scope row.
"""
from __future__ import annotations

import os
import logging
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Set, Tuple

import numpy as np

from .. import jf_io
from .._native import KdfError
from ..engine import KmerEngine, hit_positions
from ..kmer_fasta import read_kmer_fasta_keys
from ..reads import FLAG_OFF_MODULE3, bam_reader, kmers_to_keys

logger = logging.getLogger(__name__)

# stream positions per scan launch (the reference batches 5000 reads per query,
# core/bam_scanner.py:247; batch size does not change results)
SCAN_BATCH_BASES = 1 << 26
# host threads of the BAM feeder (BGZF inflate + record parsing); the reference scans contigs in a process pool
READER_THREADS = max(1, min(8, os.cpu_count() or 1))

_worker_engine: Optional[KmerEngine] = None
_worker_kmer_size: Optional[int] = None
_worker_min_distinct_kmers_per_read = 1


@dataclass
class InformativeRead:
    """One scanned record that passed the distinct-k-mer threshold."""
    query_name: str
    flag: int
    ref_id: int
    pos: int
    kmer_hit_indices: np.ndarray          # query start indices of hit windows
    n_distinct: int                       # len(unique_in_read)

    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def is_supplementary(self): return bool(self.flag & 0x800)


def _init_scan_worker(proband_data, kmer_size, min_distinct_kmers_per_read=1, device: int = 0):
    """Load the proband-unique k-mers into an HBM table.  *proband_data* is an
    index path ending in ``.jf``, a k-mer FASTA path, or an iterable of canonical
    k-mer strings -- the same dispatch as the reference (:250-281)."""
    global _worker_engine, _worker_kmer_size, _worker_min_distinct_kmers_per_read
    if _worker_engine is not None:
        _worker_engine.close()
        _worker_engine = None
    try:
        if isinstance(proband_data, str) and proband_data.endswith(".jf"):
            eng = KmerEngine(kmer_size, capacity_hint=max(jf_io.index_records(proband_data), 1), device=device)
            jf_io.load_index_into(eng, proband_data, expect_k=kmer_size)
            lo = None
        elif isinstance(proband_data, str):
            lo, hi = read_kmer_fasta_keys(proband_data, kmer_size)
            cnt = np.ones(len(lo), np.uint32)
        else:
            lo, hi = kmers_to_keys(list(proband_data), kmer_size)
            cnt = np.ones(len(lo), np.uint32)
        if lo is not None:
            eng = KmerEngine(kmer_size, capacity_hint=max(len(lo), 1), device=device)
            eng.add_pairs(lo, hi, cnt)
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish query failed: {e}") from e
    _worker_engine = eng
    _worker_kmer_size = kmer_size
    _worker_min_distinct_kmers_per_read = min_distinct_kmers_per_read


def scan_bam_for_hits(child_bam, engine: Optional[KmerEngine] = None, min_dk_per_read: Optional[int] = None,
                      batch_bases: int = SCAN_BATCH_BASES) -> Iterator[Tuple[int, List[InformativeRead]]]:
    """Scan every non-SECONDARY, non-DUPLICATE record (supplementary kept, no
    QNAME collapse: reference :405-409).  Yields (reads_scanned_in_batch,
    [InformativeRead ...]) per batch, records in file order."""
    eng = engine or _worker_engine
    if eng is None:
        raise RuntimeError("scan worker not initialised (_init_scan_worker)")
    min_dk = _worker_min_distinct_kmers_per_read if min_dk_per_read is None else min_dk_per_read
    with bam_reader(child_bam, flag_off=FLAG_OFF_MODULE3, collapse=False, max_bases=batch_bases,
                    max_reads=1 << 20, threads=READER_THREADS, want_meta=True) as rd:
        for batch in rd:
            try:
                hits, distinct = eng.scan(batch)
            except KdfError as e:
                raise RuntimeError(f"jellyfish query failed: {e}") from e
            out = []
            for r in np.flatnonzero(distinct >= max(min_dk, 1)).tolist():
                s, e_ = int(batch.offsets[r]), int(batch.offsets[r + 1]) - 1
                out.append(InformativeRead(batch.name(r), int(batch.flags[r]), int(batch.ref_ids[r]),
                                           int(batch.positions[r]), hit_positions(hits, s, e_),
                                           int(distinct[r])))
            if min_dk <= 0:
                # reference: `len(unique_in_read) < min_dk_per_read` never true for 0 -> every read kept
                keep = set(np.flatnonzero(distinct == 0).tolist())
                for r in sorted(keep):
                    out.append(InformativeRead(batch.name(r), int(batch.flags[r]), int(batch.ref_ids[r]),
                                               int(batch.positions[r]), np.zeros(0, np.int64), 0))
            yield batch.n_reads, out


def count_informative_reads(child_bam, engine: Optional[KmerEngine] = None,
                            min_dk_per_read: Optional[int] = None):
    """(total_informative, unmapped_informative, total_reads_scanned, informative records)
    with the reference's de-duplication: one task per contig plus one for
    unplaced reads, records de-duplicated by (query_name, is_supplementary)
    inside a task (core/bam_scanner.py:293-299) and again when tasks are merged
    (discovery/pipeline.py:840-846); unmapped informative reads are counted per
    task."""
    tasks = {}
    scanned = 0
    order = []
    for n, infos in scan_bam_for_hits(child_bam, engine, min_dk_per_read):
        scanned += n
        for inf in infos:
            if inf.ref_id not in tasks:
                tasks[inf.ref_id] = []
                order.append(inf.ref_id)
            tasks[inf.ref_id].append(inf)
    seen_global: Set[Tuple[str, bool]] = set()
    kept: List[InformativeRead] = []
    unmapped = 0
    for ref_id in sorted(t for t in order if t >= 0) + ([-1] if -1 in tasks else []):
        seen_local: Set[Tuple[str, bool]] = set()
        for inf in tasks[ref_id]:
            key = (inf.query_name, inf.is_supplementary)
            if key in seen_local:
                continue
            seen_local.add(key)
            if inf.is_unmapped:
                unmapped += 1
                continue
            if key in seen_global:
                continue
            kept.append(inf)
        seen_global |= seen_local
    return len(kept) + unmapped, unmapped, scanned, kept


# ---------------------------------------------------------------------------
# N1: hits -> reference coordinates, per-read SV metadata (host post-processing
# of the scan kernel's output; reference :54-117, :284-337, :340-392)
# ---------------------------------------------------------------------------

_REF_CONSUMING = (0, 2, 3, 7, 8)     # M D N = X
_ALIGNED = (0, 7, 8)                 # M = X  (pysam get_aligned_pairs(matches_only=True))
_QUERY_CONSUMING = (0, 1, 4, 7, 8)   # M I S = X


def _extract_softclips(cigartuples):
    """(left, right) soft-clip lengths; hard clips outside them are skipped and a
    read that is one single soft clip counts on the left only (reference :54-94)."""
    if not cigartuples:
        return (0, 0)
    core = [(op, ln) for op, ln in cigartuples if op != 5]
    if not core:
        return (0, 0)
    left = core[0][1] if core[0][0] == 4 else 0
    right = core[-1][1] if core[-1][0] == 4 else 0
    if len(core) == 1 and core[0][0] == 4:
        right = 0
    return (left, right)


def reference_end(pos, cigartuples):
    """pysam's ``reference_end``: pos + reference bases consumed."""
    return pos + sum(ln for op, ln in cigartuples if op in _REF_CONSUMING)


def _query_to_ref(pos, cigartuples, query_len):
    """int64[query_len]: reference position aligned to each query base, -1 when
    the base is inserted / clipped (only M, = and X pair bases up)."""
    q2r = np.full(query_len, -1, dtype=np.int64)
    q, r = 0, pos
    for op, ln in cigartuples:
        if op in _ALIGNED:
            q2r[q:q + ln] = np.arange(r, r + ln)
            q += ln
            r += ln
        elif op in (1, 4):
            q += ln
        elif op in (2, 3):
            r += ln
    return q2r


def _collect_kmer_ref_positions(pos, cigartuples, query_len, kmer_hit_indices, kmer_size):
    """Counter {reference position: number of hit k-mers covering it} for one
    read (reference :97-117)."""
    import collections
    cov = collections.Counter()
    if len(kmer_hit_indices) == 0:
        return cov
    q2r = _query_to_ref(pos, cigartuples, query_len)
    qpos = (np.asarray(kmer_hit_indices, dtype=np.int64)[:, None] + np.arange(kmer_size)[None, :]).ravel()
    qpos = qpos[qpos < query_len]
    rpos = q2r[qpos]
    rpos = rpos[rpos >= 0]
    if len(rpos):
        u, c = np.unique(rpos, return_counts=True)
        cov.update(dict(zip(u.tolist(), c.tolist())))
    return cov


def _decode_read(batch, r):
    """Upper-case bases of record r of a batch (N for invalid positions)."""
    s, e = int(batch.offsets[r]), int(batch.offsets[r + 1]) - 1
    idx = np.arange(s, e)
    codes = (batch.packed[idx >> 5] >> ((idx & 31) * 2).astype(np.uint64)) & np.uint64(3)
    inv = (batch.invalid[idx >> 6] >> (idx & 63).astype(np.uint64)) & np.uint64(1)
    chars = np.frombuffer(b"ACGT", np.uint8)[codes.astype(np.intp)].copy()
    chars[inv.astype(bool)] = ord("N")
    return chars.tobytes().decode()


def scan_bam_module3(child_bam, kmer_size=None, min_dk_per_read=None, engine=None,
                     batch_bases: int = SCAN_BATCH_BASES):
    """Module 3 over a whole BAM with the reference's task structure: one task per
    contig (records whose ref_id is that contig, unmapped mates included) plus one
    for unplaced reads, each de-duplicating by (query_name, is_supplementary);
    task results merged in contig order (core/bam_scanner.py:340-507,
    discovery/pipeline.py:733-860).

    Returns the reference's tuple
      (read_hits, reads_seen, unmapped_informative, total_reads_scanned,
       read_sv_meta, kmer_coverage, read_coverage)
    with read_hits = [(ref_name, ref_start, ref_end, query_name, unique_in_read, is_supplementary)].
    """
    import collections
    from ..kmer_utils import canonicalize
    eng = engine or _worker_engine
    if eng is None:
        raise RuntimeError("scan worker not initialised (_init_scan_worker)")
    k = kmer_size or _worker_kmer_size or eng.k
    min_dk = _worker_min_distinct_kmers_per_read if min_dk_per_read is None else min_dk_per_read

    per_task = collections.OrderedDict()        # ref_id -> list of informative record dicts, file order
    total_scanned = 0
    rd = bam_reader(child_bam, flag_off=FLAG_OFF_MODULE3, collapse=False, max_bases=batch_bases,
                    max_reads=1 << 20, threads=READER_THREADS, want_aux=True)
    refs = rd.references()
    with rd:
        for batch in rd:
            total_scanned += batch.n_reads
            try:
                hits, distinct = eng.scan(batch)
            except KdfError as e:
                raise RuntimeError(f"jellyfish query failed: {e}") from e
            keep = np.flatnonzero(distinct >= min_dk) if min_dk > 0 else np.arange(batch.n_reads)
            for r in keep.tolist():
                s, e_ = int(batch.offsets[r]), int(batch.offsets[r + 1]) - 1
                idx = hit_positions(hits, s, e_)
                seq = _decode_read(batch, r)
                rec = {
                    "name": batch.name(r), "flag": int(batch.flags[r]), "ref_id": int(batch.ref_ids[r]),
                    "pos": int(batch.positions[r]), "cigar": batch.cigartuples(r), "sa": batch.sa_tag(r),
                    "qlen": len(seq), "hit_idx": idx,
                    "kmers": {canonicalize(seq[p:p + k]) for p in idx.tolist()},
                }
                per_task.setdefault(rec["ref_id"], []).append(rec)

    read_hits, reads_seen, read_sv_meta = [], set(), {}
    kmer_coverage = collections.defaultdict(collections.Counter)
    read_coverage = collections.defaultdict(collections.Counter)
    unmapped_informative = 0
    order = sorted(t for t in per_task if t >= 0) + ([-1] if -1 in per_task else [])
    for ref_id in order:
        seen_local = set()
        for rec in per_task[ref_id]:
            is_supp = bool(rec["flag"] & 0x800)
            key = (rec["name"], is_supp)
            if key in seen_local:                       # _process_informative_read: already seen in this task
                continue
            seen_local.add(key)
            if rec["flag"] & 0x4:                       # unmapped: counted, no coordinates
                unmapped_informative += 1
                continue
            chrom = refs[rec["ref_id"]]
            cov = _collect_kmer_ref_positions(rec["pos"], rec["cigar"], rec["qlen"], rec["hit_idx"], k)
            kmer_coverage[chrom].update(cov)
            for p in cov:
                read_coverage[chrom][p] += 1
            paired = bool(rec["flag"] & 0x1)
            if key not in read_sv_meta:
                read_sv_meta[key] = {
                    "has_sa": rec["sa"] is not None,
                    "sa_str": rec["sa"] if (rec["sa"] is not None and not is_supp) else None,
                    "is_paired": paired,
                    "is_proper_pair": bool(rec["flag"] & 0x2),
                    "mate_is_unmapped": bool(rec["flag"] & 0x8) if paired else False,
                    "max_clip": max([ln for op, ln in rec["cigar"] if op == 4], default=0),
                }
            if key in reads_seen:                       # already reported by an earlier task
                continue
            read_hits.append((chrom, rec["pos"], reference_end(rec["pos"], rec["cigar"]), rec["name"],
                              rec["kmers"], is_supp))
        reads_seen |= seen_local
    return (read_hits, reads_seen, unmapped_informative, total_scanned, read_sv_meta,
            kmer_coverage, read_coverage)
