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
            k, lo, hi, cnt = jf_io.read_index(proband_data, expect_k=kmer_size)
        elif isinstance(proband_data, str):
            lo, hi = read_kmer_fasta_keys(proband_data, kmer_size)
            cnt = np.ones(len(lo), np.uint32)
        else:
            lo, hi = kmers_to_keys(list(proband_data), kmer_size)
            cnt = np.ones(len(lo), np.uint32)
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
                    max_reads=1 << 20, want_meta=True) as rd:
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
