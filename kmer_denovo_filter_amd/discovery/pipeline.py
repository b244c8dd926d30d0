"""Drop-in mirror of the hot-path helpers of
``kmer_denovo_filter/discovery/pipeline.py`` (Modules 1-2):

    _extract_child_kmers_discovery   reference :69-268
    _subtract_reference_kmers        reference :271-319
    _count_parent_jellyfish          reference :322-459
    _filter_parents_discovery        reference :462-612

Same names, arguments, return values, intermediate files
(``child_candidates.fa`` -> ``child_non_ref_kmers.fa`` -> ``after_mother.fa`` ->
``proband_unique.fa``, each ``>{i}\\n{KMER}\\n``) and RuntimeError convention.
The orchestration above these helpers (run_discovery_pipeline, clustering,
writers) is out of scope and unchanged.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np

from .. import jf_io
from .._native import KdfError
from ..core.jellyfish_wrappers import (
    _engine_capacity_hint,
    _estimate_jf_hash_size,
    _format_elapsed,
    _format_file_size,
    _stream_bam,
)
from ..engine import KmerEngine
from ..kmer_fasta import read_kmer_fasta_keys, remove_with_sidecar, write_kmer_fasta

logger = logging.getLogger(__name__)


def _child_key_parts(child_bam, device=0):
    """Slices of the key space for the child count: ``KDF_KEY_PARTS`` when set, else from the BAM size and
    the free HBM.  Rule of thumb (30x human WGS: ~0.6 BAM bytes per base, ~0.14 distinct 31-mers per base with
    0.5 % errors): distinct ~ 0.23 x BAM bytes, table bytes ~ 12 x distinct / 0.6 ~ 4.6 x BAM bytes; the table
    may take 70 % of what is free (the rest is partition scratch and growth).  Small inputs give 1."""
    env = os.environ.get("KDF_KEY_PARTS")
    if env:
        return max(1, int(env))
    from ctypes import byref, c_uint64
    from .. import _native
    free, total = c_uint64(0), c_uint64(0)
    _native.check(_native.load().kdf_device_memory(device, byref(free), byref(total)))
    need = 4.6 * os.path.getsize(child_bam)
    return max(1, int(-(-need // max(1.0, 0.7 * free.value))))


def _extract_child_kmers_discovery(child_bam, ref_fasta, kmer_size, min_child_count, threads, tmpdir,
                                   jf_hash_size=None):
    """Module 1: count every canonical child k-mer, keep count >= min_child_count.

    Returns (child_candidates_fa, n_candidates)."""
    if jf_hash_size is None:
        jf_hash_size = _estimate_jf_hash_size(child_bam, kmer_size, default="1G")
    logger.info("Extracting child k-mers from BAM (k=%d, jf hash size=%s)…", kmer_size, jf_hash_size)
    extract_start = time.monotonic()
    child_candidates_fa = os.path.join(tmpdir, "child_candidates.fa")
    # A 30x human sample has ~10^10 distinct 31-mers (sequencing errors included): more than one table in
    # 288 GB of HBM holds.  KDF_KEY_PARTS = P counts the key space in P slices, one pass over the BAM each
    # (Jellyfish's answer to the same problem is to spill and merge hash files, jellyfish_wrappers.py:335-366).
    parts = _child_key_parts(child_bam)
    try:
        with KmerEngine(kmer_size, capacity_hint=max(1, _engine_capacity_hint(jf_hash_size, child_bam) // parts)) as eng:
            los, his = [], []
            if parts > 1:
                eng.set_option("key_parts", parts)
            for part in range(parts):
                if parts > 1:
                    eng.clear(); eng.set_option("key_part", part)
                _stream_bam(eng, child_bam, ref_fasta, threads, filtered=False)
                cap, distinct, windows = eng.stats()
                logger.info("Child k-mer counting complete (%s, slice %d of %d, %d windows, %d distinct, table %d slots)",
                            _format_elapsed(time.monotonic() - extract_start), part + 1, parts, windows, distinct, cap)
                logger.info("Dumping child k-mers with count >= %d…", min_child_count)
                dump_start = time.monotonic()
                lo, hi, _ = eng.export_ge(min_child_count)
                los.append(lo); his.append(hi)
            lo, hi = (np.concatenate(los), np.concatenate(his)) if parts > 1 else (los[0], his[0])
    except KdfError as e:
        raise RuntimeError(f"jellyfish count (child) failed: {e}") from e
    n_candidates = write_kmer_fasta(child_candidates_fa, lo, hi, kmer_size)
    logger.info("Child k-mer dump complete (%s, %d candidates, FASTA: %s)",
                _format_elapsed(time.monotonic() - dump_start), n_candidates,
                _format_file_size(child_candidates_fa))
    logger.info("Child candidate k-mers (count >= %d): %d", min_child_count, n_candidates)
    return child_candidates_fa, n_candidates


def _index_k(ref_jf):
    header, _ = jf_io.read_header(ref_jf)
    return int(header["key_len"]) // 2


def _subtract_reference_kmers(ref_jf, child_candidates_fa, tmpdir):
    """Keep the candidates whose count in the reference index is 0
    (``jellyfish query ref.jf -s candidates.fa`` + ``== "0"``).  Deletes the
    input FASTA.  Returns (child_non_ref_fa, n_non_ref)."""
    child_non_ref_fa = os.path.join(tmpdir, "child_non_ref_kmers.fa")
    try:
        k, rlo, rhi, rcnt = jf_io.read_index(ref_jf)
        lo, hi = read_kmer_fasta_keys(child_candidates_fa, k)
        if len(lo):
            with KmerEngine(k, capacity_hint=max(len(rlo), 1)) as eng:
                eng.add_pairs(rlo, rhi, rcnt)
                c = eng.query(lo, hi)
            keep = c == 0
            lo, hi = lo[keep], hi[keep]
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish query (ref subtraction) failed: {e}") from e
    n_non_ref = write_kmer_fasta(child_non_ref_fa, lo, hi, k)
    remove_with_sidecar(child_candidates_fa)
    logger.info("Non-reference child k-mers after subtraction: %d", n_non_ref)
    return child_non_ref_fa, n_non_ref


def _count_parent_jellyfish(parent_bam, ref_fasta, kmer_fasta, kmer_size, parent_dir, threads,
                            label="Parent", n_filter_kmers=None):
    """``samtools fasta | jellyfish count -C --if kmer_fasta`` -> index path.
    The index holds every filter k-mer (count 0 when never seen)."""
    os.makedirs(parent_dir, exist_ok=True)
    jf_output = os.path.join(parent_dir, "parent.jf")
    logger.info("%s: scanning BAM (%s): %s", label, _format_file_size(parent_bam), parent_bam)
    scan_start = time.monotonic()
    try:
        lo, hi = read_kmer_fasta_keys(kmer_fasta, kmer_size)
        logger.info("  BAM stream -> MI355X count --if (k=%d, threads=%d, filter_kmers=%d)",
                    kmer_size, threads, len(lo))
        with KmerEngine(kmer_size, capacity_hint=max(len(lo), 1)) as eng:
            eng.load_filter(lo, hi)
            _stream_bam(eng, parent_bam, ref_fasta, threads, filtered=True)
            flo, fhi, fcnt = eng.export_ge(0)
        jf_io.write_index(jf_output, kmer_size, flo, fhi, fcnt,
                          cmdline=["count", "-m", str(kmer_size), "-C", "--if", kmer_fasta, "-o", jf_output])
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish count ({label}) failed: {e}") from e
    logger.info("  %s jellyfish counting complete (%s, index: %s)", label,
                _format_elapsed(time.monotonic() - scan_start), _format_file_size(jf_output))
    return jf_output


def _query_index(jf_path, lo, hi, k, what):
    """``jellyfish query jf -s kmers.fa``: counts in input order."""
    try:
        _, ilo, ihi, icnt = jf_io.read_index(jf_path, expect_k=k)
        with KmerEngine(k, capacity_hint=max(len(ilo), 1)) as eng:
            eng.add_pairs(ilo, ihi, icnt)
            return eng.query(lo, hi)
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish query ({what}) failed: {e}") from e


def _filter_parents_discovery(mother_bam, father_bam, ref_fasta, child_non_ref_fa, kmer_size, threads, tmpdir,
                              parent_max_count=0):
    """Module 2: mother then father (on the survivors), keep
    ``count <= parent_max_count``.  Returns (n_proband_unique, path | None)."""
    lo, hi = read_kmer_fasta_keys(child_non_ref_fa, kmer_size)
    n_input = len(lo)
    if n_input == 0:
        return 0, None
    logger.info("Filtering %d non-reference k-mers against parents…", n_input)

    mother_jf = _count_parent_jellyfish(mother_bam, ref_fasta, child_non_ref_fa, kmer_size,
                                        os.path.join(tmpdir, "mother"), threads, label="Mother",
                                        n_filter_kmers=n_input)
    after_mother_fa = os.path.join(tmpdir, "after_mother.fa")
    c = _query_index(mother_jf, lo, hi, kmer_size, "mother filter")
    keep = c <= parent_max_count
    n_surviving = write_kmer_fasta(after_mother_fa, lo[keep], hi[keep], kmer_size)
    n_removed_mother = n_input - n_surviving
    if os.path.exists(mother_jf):
        os.remove(mother_jf)
    logger.info("Mother: %d / %d non-ref k-mers found (count > %d), %d surviving",
                n_removed_mother, n_input, parent_max_count, n_surviving)
    if n_surviving == 0:
        return 0, None

    lo, hi = lo[keep], hi[keep]
    father_jf = _count_parent_jellyfish(father_bam, ref_fasta, after_mother_fa, kmer_size,
                                        os.path.join(tmpdir, "father"), threads, label="Father",
                                        n_filter_kmers=n_surviving)
    proband_unique_fa = os.path.join(tmpdir, "proband_unique.fa")
    c = _query_index(father_jf, lo, hi, kmer_size, "father filter")
    keep = c <= parent_max_count
    n_proband = write_kmer_fasta(proband_unique_fa, lo[keep], hi[keep], kmer_size)
    n_removed_father = n_surviving - n_proband
    if os.path.exists(father_jf):
        os.remove(father_jf)
    remove_with_sidecar(after_mother_fa)
    logger.info("Father: %d / %d surviving k-mers found (count > %d), %d proband-unique",
                n_removed_father, n_surviving, parent_max_count, n_proband)
    logger.info("Proband-unique k-mers (absent from both parents): %d / %d", n_proband, n_input)
    logger.info("Proband-unique FASTA: %s (%s)", proband_unique_fa, _format_file_size(proband_unique_fa))
    return n_proband, proband_unique_fa


def _write_informative_reads_discovery(child_bam, ref_fasta, proband_unique_kmers_or_path, kmer_size, output_bam,
                                       threads=4):
    """Child records carrying a proband-unique k-mer -> sorted, indexed BAM with
    ``dk:i:1`` on every record (reference :1979-2079).  Same selection: secondary
    and duplicate records skipped, unmapped and low-MAPQ ones kept, first record
    per (query name, is_supplementary).  The probe is the engine's scan kernel
    (the reference dispatches to ``jellyfish query`` or an Aho-Corasick automaton
    on the same three input kinds); the copy, sort and index are
    ``kdf_bam_write_subset`` instead of pysam / ``samtools sort`` / ``samtools index``.
    Returns the number of records written."""
    from ..core import bam_scanner
    from ..reads import bam_reader, write_bam_subset
    bam_scanner._init_scan_worker(proband_unique_kmers_or_path or set(), kmer_size)
    eng = bam_scanner._worker_engine
    ordinals, written = [], set()
    try:
        with bam_reader(child_bam, flag_off=bam_scanner.FLAG_OFF_MODULE3, collapse=False,
                        max_bases=bam_scanner.SCAN_BATCH_BASES, max_reads=1 << 20, threads=threads,
                        want_meta=True) as rd:
            for batch in rd:
                _hits, distinct = eng.scan(batch)
                for r in np.flatnonzero(distinct >= 1).tolist():
                    key = (batch.name(r), bool(int(batch.flags[r]) & 0x800))
                    if key not in written:
                        written.add(key)
                        ordinals.append(int(batch.ordinals[r]))
        n = write_bam_subset(child_bam, output_bam, ordinals, [b"dkC\x01"] * len(ordinals), sort_and_index=True,
                             threads=threads)
    except KdfError as e:
        raise RuntimeError(f"informative reads BAM failed: {e}") from e
    logger.info("Informative reads BAM written: %s (%d reads)", output_bam, n)
    return n
