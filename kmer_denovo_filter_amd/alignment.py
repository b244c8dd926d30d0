"""A light alignment record with the handful of pysam.AlignedSegment members the
reference's per-read helpers use (query_sequence, query_qualities, cigartuples,
reference_start/_end, get_aligned_pairs, get_reference_positions, flags), built
from the engine's own BAM reader -- so the VCF-mode producer and Module 3's
post-processing run without pysam."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from .core.bam_scanner import _decode_read, reference_end


@dataclass
class AlignedRead:
    query_name: str
    flag: int
    reference_id: int
    reference_name: Optional[str]
    reference_start: int
    mapping_quality: int
    cigartuples: List[Tuple[int, int]]
    query_sequence: str
    query_qualities: Optional[np.ndarray]

    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def is_secondary(self): return bool(self.flag & 0x100)
    @property
    def is_supplementary(self): return bool(self.flag & 0x800)
    @property
    def is_duplicate(self): return bool(self.flag & 0x400)
    @property
    def reference_end(self): return reference_end(self.reference_start, self.cigartuples)

    def get_aligned_pairs(self, matches_only: bool = False):
        """[(query_pos | None, ref_pos | None)] as pysam: M/=/X pair bases, I/S give
        (q, None), D/N give (None, r); hard clips and padding give nothing."""
        out = []
        q, r = 0, self.reference_start
        for op, ln in self.cigartuples:
            if op in (0, 7, 8):
                out.extend(zip(range(q, q + ln), range(r, r + ln)))
                q += ln; r += ln
            elif op in (1, 4):
                if not matches_only:
                    out.extend((i, None) for i in range(q, q + ln))
                q += ln
            elif op in (2, 3):
                if not matches_only:
                    out.extend((None, i) for i in range(r, r + ln))
                r += ln
        return out

    def get_reference_positions(self, full_length: bool = False):
        """Reference position of every query base (None for inserted / clipped
        bases) when full_length, else the aligned positions only."""
        pos = []
        r = self.reference_start
        for op, ln in self.cigartuples:
            if op in (0, 7, 8):
                pos.extend(range(r, r + ln)); r += ln
            elif op in (1, 4):
                if full_length:
                    pos.extend([None] * ln)
            elif op in (2, 3):
                r += ln
        return pos


def reads_from_batch(batch, refs, indices=None):
    """AlignedRead objects for the records of a batch read with want_aux=True."""
    idx = range(batch.n_reads) if indices is None else indices
    out = []
    for i in idx:
        rid = int(batch.ref_ids[i])
        out.append(AlignedRead(
            query_name=batch.name(i), flag=int(batch.flags[i]), reference_id=rid,
            reference_name=refs[rid] if 0 <= rid < len(refs) else None,
            reference_start=int(batch.positions[i]), mapping_quality=int(batch.mapq[i]),
            cigartuples=batch.cigartuples(i), query_sequence=_decode_read(batch, i),
            query_qualities=batch.qualities(i)))
    return out
