"""Drop-in mirror of the k-mer part of ``kmer_denovo_filter/kmer_utils.py``
(reference lines 15-38, 91-245): ``reverse_complement``, ``canonicalize``,
``_extract_read_kmers`` and ``JellyfishKmerQuery``.

The string helpers are kept for API compatibility (callers use them on single
k-mers); bulk work goes through the engine.  ``JellyfishKmerQuery`` keeps its
duck type (``query_batch``, ``scan_read``, ``close``, ``_cache``, ``jf_path``) but
probes an HBM-resident table instead of spawning ``jellyfish query`` per batch.
The GPU context is created on first use, never at construction: the reference
builds these objects inside forked workers (core/bam_scanner.py:250-281).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import jf_io
from ._native import KdfError
from .engine import KmerEngine, hit_positions
from .reads import ReadStream, keys_to_kmers, kmers_to_keys

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def reverse_complement(seq):
    """Reverse complement (case preserved), reference :30-32."""
    return seq[::-1].translate(_COMP)


def canonicalize(kmer):
    """Lexicographically smaller of a k-mer and its reverse complement, reference :35-38."""
    return min(kmer, reverse_complement(kmer))


def _is_symbolic(allele):
    """True for VCF alleles without a literal sequence: ``<DEL>``-style, breakends,
    ``*`` and the empty allele (reference :18-27)."""
    if not allele:
        return True
    return allele[0] == "<" or allele == "*" or "[" in allele or "]" in allele


def read_supports_alt(read, variant_pos, ref, alt, min_baseq=0, *, aligned_pairs=None, seq=None, quals=None):
    """True when the read bases aligned to the reference span of the variant
    spell exactly *alt* (reference :1037-1101).  A base below *min_baseq* inside
    the span disqualifies the read; symbolic / missing ALT never match."""
    if alt is None or _is_symbolic(alt):
        return False
    seq = read.query_sequence if seq is None else seq
    if seq is None:
        return False
    if min_baseq > 0 and quals is None:
        quals = read.query_qualities
    pairs = read.get_aligned_pairs(matches_only=False) if aligned_pairs is None else aligned_pairs
    span_end = variant_pos + len(ref)
    started, bases = False, []
    for qpos, rpos in pairs:
        if rpos is not None and rpos >= span_end:
            break
        if rpos == variant_pos:
            started = True
        if started and qpos is not None:
            if min_baseq > 0 and quals is not None and quals[qpos] < min_baseq:
                return False
            bases.append(seq[qpos])
    if not started:
        return False
    return "".join(bases).upper() == alt.upper()


def extract_variant_spanning_kmers(read, variant_pos, k, min_baseq=0, ref=None, alt=None, *,
                                   aligned_pairs=None, seq=None, quals=None):
    """Canonical k-mers of the read whose window covers the variant (for an
    insertion: any of its len(alt) read bases); windows with an N or a base
    below *min_baseq* are dropped (reference :1104-1172).  The N / quality test
    looks at the upper-cased read, the k-mer itself keeps the read's case, as
    in the reference."""
    try:
        at = read.get_reference_positions(full_length=True).index(variant_pos)
    except ValueError:
        return set()
    seq = read.query_sequence if seq is None else seq
    if seq is None:
        return set()
    if quals is None:
        quals = read.query_qualities
    alt_len = len(alt) if alt and not _is_symbolic(alt) else 1
    first = max(0, at - k + 1)
    last = min(len(seq) - k, at + alt_len - 1)
    if last < first:
        return set()
    upper = seq[first:last + k].upper()
    bad = np.frombuffer(upper.encode(), dtype=np.uint8) == ord("N")
    if quals is not None and min_baseq > 0:
        bad = bad | (np.asarray(quals[first:last + k]) < min_baseq)
    csum = np.concatenate(([0], np.cumsum(bad)))
    kmers = set()
    for s0 in range(first, last + 1):
        o = s0 - first
        if csum[o + k] - csum[o] == 0:
            kmers.add(canonicalize(seq[s0:s0 + k]))
    return kmers


def _extract_read_kmers(seq, kmer_size):
    """(canon_at_pos, unique_candidates) as the reference builds them (:91-121): the read is upper-cased, a
    window holding an 'N' yields nothing, candidates keep first-seen order.  The N test is one running sum
    over the read instead of a substring search per window."""
    n_windows = len(seq) - kmer_size + 1
    if n_windows <= 0:
        return {}, []
    upper = seq.upper()
    n_before = np.concatenate(([0], np.cumsum(np.frombuffer(upper.encode(), dtype=np.uint8) == ord("N"))))
    clean = np.flatnonzero(n_before[kmer_size:] == n_before[:n_windows])          # window starts without an N
    canon_at_pos = {int(i): canonicalize(upper[i:i + kmer_size]) for i in clean.tolist()}
    return canon_at_pos, list(dict.fromkeys(canon_at_pos.values()))


class KmerAutomaton:
    """Stand-in for the ``ahocorasick.Automaton`` that ``build_kmer_automaton``
    returns in the reference (:41-65): ``iter(seq)`` yields ``(end_index,
    canonical_kmer)`` for every window of *seq* whose canonical k-mer is in the
    set, ``len()`` is the number of patterns (forward + distinct reverse
    complements).  Backed by the engine's scan kernel; the table is created on
    first use.  Unlike Aho-Corasick it matches case-insensitively (the reference's
    AC backend is case-sensitive on the raw read, SURVEY.md "Known quirks")."""

    def __init__(self, canonical_kmers, device: int = 0):
        self._kmers = list(dict.fromkeys(canonical_kmers))
        self.kmer_size = len(self._kmers[0])
        if any(len(km) != self.kmer_size for km in self._kmers):
            raise ValueError("k-mers of different lengths")
        self._n_patterns = sum(1 if reverse_complement(km) == km else 2 for km in self._kmers)
        self._device = device
        self._engine: Optional[KmerEngine] = None

    def __len__(self):
        return self._n_patterns

    def _ensure_engine(self) -> KmerEngine:
        if self._engine is None:
            lo, hi = kmers_to_keys(self._kmers, self.kmer_size, canonical=True)
            eng = KmerEngine(self.kmer_size, capacity_hint=max(len(lo), 1), device=self._device)
            eng.add_pairs(lo, hi, np.ones(len(lo), np.uint32))
            self._engine = eng
        return self._engine

    def iter(self, seq):
        k = self.kmer_size
        if len(seq) < k:
            return
        eng = self._ensure_engine()
        hits, _ = eng.scan(ReadStream.from_strings([seq]), want_distinct=False)
        up = seq.upper()
        for p in hit_positions(hits, 0, len(seq)).tolist():
            yield p + k - 1, canonicalize(up[p:p + k])

    def __del__(self):
        try:
            if self._engine is not None:
                self._engine.close()
        except Exception:  # noqa: BLE001
            pass


def build_kmer_automaton(canonical_kmers):
    """Probe structure over a set of canonical k-mers, or ``None`` when the set is
    empty (reference :41-65)."""
    kmers = list(canonical_kmers)
    if not kmers:
        return None
    return KmerAutomaton(kmers)


class JellyfishKmerQuery:
    """Membership probe of canonical k-mers against an index file (reference
    :124-245).  *jf_path* may be a ``kdf/sorted`` index written by this package or
    a real Jellyfish ``binary/sorted`` file."""

    def __init__(self, jf_path, device: int = 0):
        self.jf_path = jf_path
        self._cache: Dict[str, bool] = {}    # canonical_kmer -> present (count > 0)
        self._device = device
        self._engine: Optional[KmerEngine] = None
        self.kmer_size: Optional[int] = None

    # -- engine ------------------------------------------------------------
    def _ensure_engine(self) -> KmerEngine:
        if self._engine is None:
            try:
                k = int(jf_io.read_header(self.jf_path)[0]["key_len"]) // 2
                eng = KmerEngine(k, capacity_hint=max(jf_io.index_records(self.jf_path), 1), device=self._device)
                jf_io.load_index_into(eng, self.jf_path)
            except (KdfError, ValueError, OSError) as e:
                raise RuntimeError(f"jellyfish query failed: {e}") from e
            self._engine, self.kmer_size = eng, k
        return self._engine

    def _subprocess_query(self, kmers):
        """Present (count > 0) subset of *kmers*; name kept from the reference
        (:152-183) although no subprocess is involved."""
        kmers = list(kmers)
        if not kmers:
            return set()
        eng = self._ensure_engine()
        ok = [km for km in kmers if len(km) == eng.k and set(km.upper()) <= set("ACGT")]
        if not ok:
            return set()
        lo, hi = kmers_to_keys(ok, eng.k, canonical=True)
        try:
            counts = eng.query(lo, hi)
        except KdfError as e:
            raise RuntimeError(f"jellyfish query failed: {e}") from e
        # jellyfish prints the canonical k-mer; callers pass canonical k-mers
        return {canonicalize(km.upper()) for km, c in zip(ok, counts) if c != 0}

    def query_batch(self, canonical_kmers):
        """Set of the given canonical k-mers that are present, with a result
        cache (reference :185-207)."""
        if not canonical_kmers:
            return set()
        uncached = [k for k in canonical_kmers if k not in self._cache]
        if uncached:
            hits = self._subprocess_query(uncached)
            for k in uncached:
                self._cache[k] = k in hits
        return {k for k in canonical_kmers if self._cache.get(k, False)}

    def scan_read(self, seq, kmer_size):
        """(unique_in_read, kmer_hit_indices) for one read (reference :209-238):
        one scan-kernel launch over the read instead of a subprocess."""
        if len(seq) < kmer_size:
            return set(), set()
        eng = self._ensure_engine()
        if kmer_size != eng.k:
            raise RuntimeError(f"jellyfish query failed: index has k={eng.k}, asked for k={kmer_size}")
        st = ReadStream.from_strings([seq])
        try:
            hits, _ = eng.scan(st, want_distinct=False)
        except KdfError as e:
            raise RuntimeError(f"jellyfish query failed: {e}") from e
        pos = hit_positions(hits, 0, len(seq))
        if len(pos) == 0:
            return set(), set()
        up = seq.upper()
        unique_in_read = set()
        for p in pos.tolist():
            c = canonicalize(up[p:p + kmer_size])
            unique_in_read.add(c)
            self._cache[c] = True
        return unique_in_read, set(pos.tolist())

    def close(self):
        """Clear the result cache (reference :240-242); the table stays resident."""
        self._cache.clear()

    def release(self):
        """Free the HBM table (no reference equivalent: jellyfish's mmap dies with its process)."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def __del__(self):
        try:
            self.close()
            self.release()
        except Exception:  # noqa: BLE001
            pass
