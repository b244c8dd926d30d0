"""Drop-in mirror of ``kmer_denovo_filter/core/jellyfish_wrappers.py``.

Same function names, arguments, return values and error behaviour as the
reference, but every ``samtools fasta | jellyfish count/dump/query/merge``
subprocess is replaced by the MI355X engine (libkdf.so).  ``.jf`` paths returned
here are ``kdf/sorted`` index files (jf_io.py); a real Jellyfish ``binary/sorted``
file supplied by the user (``--ref-jf``) is read as is.

Reference lines: _find_jf_files 59-70, _estimate_jf_hash_size 73-107,
_scan_parent_jellyfish 115-283, _ensure_ref_jf 286-332, _merge_jf_files 335-366,
_build_proband_jf_index 369-436.
"""
from __future__ import annotations

import glob
import logging
import os
import time

import numpy as np

from .. import dist_env, jf_io
from .._native import KdfError
from ..engine import KmerEngine
from ..kmer_fasta import read_kmer_fasta_keys
from ..reads import bam_reader, fasta_reader, keys_to_kmers, stream_batches_overlapped

logger = logging.getLogger(__name__)

# stream positions per host->HBM batch (bounded host memory: 2 bits + 1 bit per base)
BATCH_BASES = 1 << 26


def _format_elapsed(seconds):
    """``12.3s`` / ``4m 5.0s`` / ``1h 2m 3s``: the strings of the reference's progress log."""
    minutes, sec = divmod(seconds, 60)
    if minutes < 1:
        return "%.1fs" % seconds
    hours, minutes = divmod(int(minutes), 60)
    return "%dh %dm %.0fs" % (hours, minutes, sec) if hours else "%dm %.1fs" % (minutes, sec)


_SIZE_UNITS = ("B", "KB", "MB", "GB", "TB", "PB")


def _format_file_size(path):
    """Size of a file in powers of 1024 with one decimal (``?`` when it cannot be read)."""
    try:
        size = float(os.path.getsize(path))
    except OSError:
        return "?"
    step = 0
    while size >= 1024 and step < len(_SIZE_UNITS) - 1:
        size /= 1024
        step += 1
    return "%.1f %s" % (size, _SIZE_UNITS[step])


def _find_jf_files(base_path):
    """``base.jf`` when it exists, then its overflow chunks ``base.jf_N`` in name order (reference :59-70:
    Jellyfish spills such chunks when its table is too small; the engine grows in HBM and never does, but an
    index a user built with Jellyfish may come in pieces)."""
    chunks = sorted(glob.glob(glob.escape(base_path) + "_[0-9]*"))
    return ([base_path] if os.path.exists(base_path) else []) + chunks


def _estimate_jf_hash_size(bam_path, kmer_size, default="1G"):
    """The reference's ``-s`` heuristic (:73-107): 0.3 table entries per BAM byte, held between 100 M and 4 G,
    printed in whole G from 10^9 entries on, else in whole M."""
    try:
        entries = os.path.getsize(bam_path) * 3 // 10
    except OSError:
        return default
    entries = min(max(entries, 100_000_000), 4_000_000_000)
    unit, div = ("G", 10**9) if entries >= 10**9 else ("M", 10**6)
    return "%d%s" % (entries // div, unit)


def _parse_hash_size(s, default=1 << 20):
    """Jellyfish ``-s`` strings: plain integers or K/M/G suffixes."""
    if s is None:
        return default
    if isinstance(s, (int, np.integer)):
        return int(s)
    s = str(s).strip()
    mult = {"k": 10**3, "K": 10**3, "m": 10**6, "M": 10**6, "g": 10**9, "G": 10**9}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(s)


def _engine_capacity_hint(hash_size, bam_path=None):
    """The reference sizes Jellyfish's table up front because overflow spills to
    disk.  The engine grows in HBM, so the hint only needs to be in the right
    ballpark: min(requested, 3 bits of BAM per distinct k-mer guess)."""
    want = _parse_hash_size(hash_size)
    if bam_path is not None:
        try:
            want = min(want, max(1 << 20, os.path.getsize(bam_path) * 3 // 10))
        except OSError:
            pass
    return max(1 << 16, want)


def _stream_bam(engine, bam_path, ref_fasta, threads, filtered):
    """``samtools fasta -F 0xD00 bam | jellyfish count ... /dev/fd/0``."""
    if str(bam_path).endswith(".cram"):
        raise RuntimeError(
            "jellyfish count failed: CRAM input needs htslib, which the MI355X engine does not link; "
            "convert to BAM (samtools view -b) first"
        )
    # decode (reader threads + their inflate workers) | H2D from pinned buffers on a copy stream | count: three stages
    # that overlap, where the reference's pipe overlaps samtools and jellyfish.  The file is read as BGZF RANGES cut on
    # record (QNAME-run) boundaries (kdf_bam_open_range): a rank of a multi-GPU job takes its contiguous share of them
    # -- reads are independent units, SURVEY.md section 8e -- and inside a process every range has its own inflate /
    # chunk / parse pipeline (one pipeline's chunker is a serial stage: 2.0 Gbase/s whatever the thread count).
    world, rank, _ = dist_env.world_rank()
    local = _reader_pipelines(threads)
    per = max(1, threads // local)
    readers = []
    try:
        for j in range(local):
            readers.append(bam_reader(bam_path, max_bases=BATCH_BASES, max_reads=1 << 21, threads=per,
                                      part=rank * local + j, parts=world * local))
        return stream_batches_overlapped(engine, readers, filtered)
    finally:
        for rd in readers:
            rd.close()


def _reader_pipelines(threads):
    """Reader pipelines per process for ``threads`` host threads (KDF_READER_PIPELINES overrides): one per four threads."""
    env = os.environ.get("KDF_READER_PIPELINES")
    if env:
        return max(1, int(env))
    return max(1, min(16, int(threads) // 4))


def _scan_parent_jellyfish(parent_bam, ref_fasta, kmer_fasta, kmer_size, parent_dir, threads=4,
                           n_filter_kmers=None):
    """Dict canonical k-mer -> count in the parent, for the k-mers of
    *kmer_fasta* seen at least once (reference :115-283: count -C --if, then
    ``dump -c -L 1``)."""
    os.makedirs(parent_dir, exist_ok=True)
    bam_size = _format_file_size(parent_bam)
    logger.info("Scanning parent BAM (%s): %s", bam_size, parent_bam)
    scan_start = time.monotonic()
    try:
        lo, hi = read_kmer_fasta_keys(kmer_fasta, kmer_size)
        logger.info("  BAM stream -> MI355X count --if (k=%d, threads=%d, filter_kmers=%d)",
                    kmer_size, threads, len(lo))
        with KmerEngine(kmer_size, capacity_hint=max(len(lo), 1), device=_device()) as eng:
            eng.load_filter(lo, hi)
            _stream_bam(eng, parent_bam, ref_fasta, threads, filtered=True)
            _merge_filter_counts(eng, lo, hi)                  # (several ranks: every table now holds the summed counts)
            logger.info("  Jellyfish counting complete (%s)", _format_elapsed(time.monotonic() - scan_start))
            flo, fhi, fcnt = eng.export_ge(1)
    except KdfError as e:
        raise RuntimeError(f"jellyfish count failed: {e}") from e
    kmers = keys_to_kmers(flo, fhi, kmer_size)
    return dict(zip(kmers, fcnt.tolist()))


def _device():
    """The GPU of this process: LOCAL_RANK under a launcher (one process per GPU), else 0."""
    world, rank, host = dist_env.world_rank()
    if world == 1 or host:
        return 0
    return int(os.environ.get("LOCAL_RANK", rank))


def _merge_filter_counts(eng, lo, hi, dlo=None, dhi=None):
    """After every rank counted its shard of a parent's reads against the replicated filter: ONE all-reduce(sum) of the
    per-key counts (distributed.ShardedFilterCount: the `ncclAllReduce` of SURVEY.md section 8e), and the sums go back
    into every rank's table (kdf_set_counts_dev), so that what follows -- `query`, `dump -L 1`, the index file -- is the
    same on every rank as in a one-process run.  ``lo`` / ``hi``: the filter keys as host arrays, or ``dlo`` / ``dhi``
    as device tensors.  One rank: nothing to do."""
    world, rank, host = dist_env.world_rank()
    if world == 1:
        return
    import torch
    from .. import devkeys
    from ..distributed import EngineOps, ShardedFilterCount
    dev = torch.device("cuda", eng.device)
    if dlo is None:
        if len(lo) == 0:
            return
        dlo, dhi = devkeys.from_host(lo, hi, eng.wide, eng.device)
    if dlo.numel() == 0:
        return
    merged = ShardedFilterCount(EngineOps(eng, dev), stage_through_host=host).merged_counts(dlo, dhi)
    m32 = merged.to(torch.int32).contiguous()                # (values < 2^32: the bit pattern is the uint32 count)
    torch.cuda.current_stream(dev).synchronize()
    eng.set_counts_dev(dlo.data_ptr(), dhi.data_ptr() if dhi is not None else None, m32.data_ptr(), int(dlo.numel()))
    eng.synchronize()


def _count_fasta_to_index(fasta_path, kmer_size, out_path, capacity_hint, cmdline):
    with KmerEngine(kmer_size, capacity_hint=capacity_hint) as eng:
        with fasta_reader(fasta_path, kmer_size, max_bases=BATCH_BASES) as rd:
            for batch in rd:
                eng.count(batch)
        lo, hi, cnt = eng.export_ge(0)
    jf_io.write_index_auto(out_path, kmer_size, lo, hi, cnt, cmdline=cmdline)    # (KDF_JF_FORMAT=jellyfish: a real binary/sorted file)
    return len(lo)


def _ensure_ref_jf(ref_fasta, kmer_size, threads, ref_jf=None):
    """Return the reference index path, building it when absent (reference
    :286-332).  An existing file is returned untouched -- it may be a real
    Jellyfish ``binary/sorted`` index given with ``--ref-jf``."""
    if ref_jf is None:
        ref_jf = f"{ref_fasta}.k{kmer_size}.jf"
    if os.path.isfile(ref_jf):
        logger.info("Reference Jellyfish index found: %s", ref_jf)
        return ref_jf
    logger.info("Building reference Jellyfish index: %s (k=%d, threads=%d)", ref_jf, kmer_size, threads)
    ref_hash_size = _estimate_jf_hash_size(ref_fasta, kmer_size, default="3G")
    logger.info("  Reference JF hash size: %s", ref_hash_size)
    build_start = time.monotonic()
    try:
        hint = max(1 << 16, min(_parse_hash_size(ref_hash_size), os.path.getsize(ref_fasta)))
        _count_fasta_to_index(ref_fasta, kmer_size, ref_jf, hint,
                              ["count", "-m", str(kmer_size), "-C", ref_fasta, "-o", ref_jf])
    except (KdfError, OSError) as e:
        raise RuntimeError(f"jellyfish count (reference) failed: {e}") from e
    logger.info("Reference index built in %s (%s)", _format_elapsed(time.monotonic() - build_start),
                _format_file_size(ref_jf))
    return ref_jf


def _merge_jf_files(jf_files, merged_path, threads=4):
    """Sum the counts of several index files (reference :335-366)."""
    if len(jf_files) <= 1:
        return jf_files[0] if jf_files else None
    logger.info("Merging %d Jellyfish chunks into %s…", len(jf_files), merged_path)
    merge_start = time.monotonic()
    try:
        k0, lo, hi, cnt = jf_io.read_index(jf_files[0])
        with KmerEngine(k0, capacity_hint=max(len(lo), 1)) as eng:
            eng.add_pairs(lo, hi, cnt)
            for f in jf_files[1:]:
                k1, lo, hi, cnt = jf_io.read_index(f, expect_k=k0)
                eng.add_pairs(lo, hi, cnt)
            lo, hi, cnt = eng.export_ge(0)
        jf_io.write_index(merged_path, k0, lo, hi, cnt, cmdline=["merge", "-o", merged_path, *jf_files])
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish merge failed: {e}") from e
    for f in jf_files:
        if f != merged_path and os.path.exists(f):
            os.remove(f)
    logger.info("Jellyfish merge complete (%s, merged: %s)", _format_elapsed(time.monotonic() - merge_start),
                _format_file_size(merged_path))
    return merged_path


def _build_proband_jf_index(proband_unique_fa, kmer_size, tmpdir, n_proband_unique=None):
    """Index of the proband-unique k-mers for Module 3 (reference :369-436:
    ``jellyfish count -C`` of the k-mer FASTA, so every k-mer has count >= 1)."""
    if n_proband_unique is None:
        n_proband_unique = 0
        with open(proband_unique_fa) as fh:
            for line in fh:
                if line.rstrip() and not line.startswith(">"):
                    n_proband_unique += 1
    hash_size = max(n_proband_unique * 2, 1_000_000)
    proband_jf = os.path.join(tmpdir, "proband_unique.jf")
    logger.info("Building Jellyfish index from %d proband-unique k-mers (hash size: %s)…",
                n_proband_unique, hash_size)
    build_start = time.monotonic()
    try:
        _count_fasta_to_index(proband_unique_fa, kmer_size, proband_jf, max(n_proband_unique, 1024),
                              ["count", "-m", str(kmer_size), "-C", "-o", proband_jf, proband_unique_fa])
    except (KdfError, OSError) as e:
        raise RuntimeError(f"jellyfish count (proband index) failed: {e}") from e
    logger.info("Proband Jellyfish index built (%s, index: %s)", _format_elapsed(time.monotonic() - build_start),
                _format_file_size(proband_jf))
    return proband_jf
