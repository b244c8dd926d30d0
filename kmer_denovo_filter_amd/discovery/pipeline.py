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

from .. import devkeys, dist_env, jf_io
from .._native import KdfError
from ..core.jellyfish_wrappers import (
    _device,
    _engine_capacity_hint,
    _estimate_jf_hash_size,
    _format_elapsed,
    _format_file_size,
    _merge_filter_counts,
    _stream_bam,
)
from ..engine import KmerEngine
from ..kmer_fasta import read_kmer_fasta_keys, remove_with_sidecar, write_kmer_fasta

logger = logging.getLogger(__name__)


def _child_key_parts(child_bam, device=0, world=1):
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
    need = 4.6 * os.path.getsize(child_bam) / max(1, world)      # (several ranks: every rank holds its share of the keys twice -- local + owned)
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
    world, rank, host = dist_env.world_rank()
    parts = _child_key_parts(child_bam, _device(), world)
    owner_eng = merger = None
    try:
        local_hint = max(1, _engine_capacity_hint(jf_hash_size, child_bam) // parts)
        with KmerEngine(kmer_size, capacity_hint=max(1, local_hint // world) if world > 1 else local_hint, device=_device()) as eng:
            if world > 1:
                # one process per GPU: every rank counts its ranges of the BAM into a local table, one owner-partitioned
                # exchange moves each (k-mer, count) pair to the rank that owns the k-mer, the owner sums -- and `dump -L`
                # runs on the owners' tables (distributed.OwnerPartitionedCount; SURVEY.md section 8e "full count stage")
                import torch
                from ..distributed import EngineOps, OwnerPartitionedCount
                dev = torch.device("cuda", eng.device)
                owner_eng = KmerEngine(kmer_size, capacity_hint=max(1, local_hint // world), device=eng.device)
                merger = OwnerPartitionedCount(EngineOps(eng, dev), device=dev, owner_ops=EngineOps(owner_eng, dev), stage_through_host=host)
            los, his = [], []
            dev_sets = []
            if parts > 1:
                eng.set_option("key_parts", parts)
            for part in range(parts):
                if parts > 1:
                    eng.clear(); eng.set_option("key_part", part)
                    if owner_eng is not None:
                        owner_eng.clear()
                _stream_bam(eng, child_bam, ref_fasta, threads, filtered=False)
                cap, distinct, windows = eng.stats()
                logger.info("Child k-mer counting complete (%s, slice %d of %d, %d windows, %d distinct, table %d slots)",
                            _format_elapsed(time.monotonic() - extract_start), part + 1, parts, windows, distinct, cap)
                logger.info("Dumping child k-mers with count >= %d…", min_child_count)
                dump_start = time.monotonic()
                # the dump stays in HBM for the next stage (ascending keys: Jellyfish's dump order is not reproducible and
                # nothing downstream relies on it, but the contract files are then the same bytes on every run)
                if merger is None:
                    dlo, dhi = devkeys.dump_ge(eng, min_child_count, eng.device)
                else:
                    merger.exchange()
                    dlo, dhi = devkeys.dump_ge(owner_eng, min_child_count, eng.device)
                    dlo, dhi = dist_env.all_gather_keys(dlo, dhi)           # every rank holds the whole candidate set
                dev_sets.append((dlo, dhi))
                lo, hi = devkeys.to_host(dlo, dhi)
                los.append(lo); his.append(hi)
            lo, hi = (np.concatenate(los), np.concatenate(his)) if parts > 1 else (los[0], his[0])
            if parts > 1:
                import torch
                dev_sets = [(torch.cat([d[0] for d in dev_sets]), torch.cat([d[1] for d in dev_sets]) if eng.wide else None)]
    except KdfError as e:
        raise RuntimeError(f"jellyfish count (child) failed: {e}") from e
    finally:
        if owner_eng is not None:
            owner_eng.close()
    if world > 1:                                                  # (rank order of the gathered sets is not key order)
        order = np.lexsort((lo, hi))
        lo, hi = lo[order], hi[order]
        import torch
        o = torch.from_numpy(order).to(dev_sets[0][0].device)
        dev_sets = [(dev_sets[0][0][o].contiguous(), dev_sets[0][1][o].contiguous() if dev_sets[0][1] is not None else None)]
    n_candidates = len(lo)
    if rank == 0:
        write_kmer_fasta(child_candidates_fa, lo, hi, kmer_size)
    dist_env.barrier()                                             # the file exists (and has its final size) on every rank
    devkeys.register(child_candidates_fa, dev_sets[0][0], dev_sets[0][1], kmer_size)
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
        k = _index_k(ref_jf)
        dev = devkeys.lookup(child_candidates_fa, k)             # the candidates are still in HBM when Module 1 ran here
        if dev is None:
            lo, hi = read_kmer_fasta_keys(child_candidates_fa, k)
            dev = devkeys.from_host(lo, hi, k > 32) if len(lo) else None
        world, rank, host = dist_env.world_rank()
        if dev is not None and dev[0].numel():
            with KmerEngine(k, capacity_hint=max(jf_io.index_records(ref_jf) // world, 1), device=_device()) as eng:
                # memory-mapped, block by block (a human index is 30 GB); several ranks: every rank loads ITS share of the
                # index's records and answers for all candidates, one all-reduce(sum) gives `jellyfish query`'s counts
                jf_io.load_index_into(eng, ref_jf, part=rank, parts=world)
                if world == 1:
                    keep = devkeys.query(eng, dev[0], dev[1], eng.device) == 0
                else:
                    import torch
                    from ..distributed import EngineOps, ShardedFilterCount
                    keep = ShardedFilterCount(EngineOps(eng, torch.device("cuda", eng.device)), stage_through_host=host).merged_counts(dev[0], dev[1]) == 0
            dlo, dhi = dev[0][keep].contiguous(), (dev[1][keep].contiguous() if dev[1] is not None else None)
            lo, hi = devkeys.to_host(dlo, dhi)
        else:
            dlo = dhi = None
            lo = hi = np.zeros(0, np.uint64)
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish query (ref subtraction) failed: {e}") from e
    n_non_ref = len(lo)
    dist_env.barrier()                                             # every rank has read what it needs of the input file
    if dist_env.is_root():
        write_kmer_fasta(child_non_ref_fa, lo, hi, k)
        remove_with_sidecar(child_candidates_fa)
    dist_env.barrier()
    if dlo is not None:
        devkeys.register(child_non_ref_fa, dlo, dhi, k)
    devkeys.forget(child_candidates_fa)
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
        with KmerEngine(kmer_size, capacity_hint=max(len(lo), 1), device=_device()) as eng:
            eng.load_filter(lo, hi)
            _stream_bam(eng, parent_bam, ref_fasta, threads, filtered=True)
            _merge_filter_counts(eng, lo, hi)                  # (several ranks: the sum of their shards' counts)
            flo, fhi, fcnt = eng.export_ge(0)
        if dist_env.is_root():
            jf_io.write_index(jf_output, kmer_size, flo, fhi, fcnt,
                              cmdline=["count", "-m", str(kmer_size), "-C", "--if", kmer_fasta, "-o", jf_output])
        dist_env.barrier()
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish count ({label}) failed: {e}") from e
    logger.info("  %s jellyfish counting complete (%s, index: %s)", label,
                _format_elapsed(time.monotonic() - scan_start), _format_file_size(jf_output))
    return jf_output


def _query_index(jf_path, lo, hi, k, what):
    """``jellyfish query jf -s kmers.fa``: counts in input order."""
    try:
        with KmerEngine(k, capacity_hint=max(jf_io.index_records(jf_path), 1)) as eng:
            jf_io.load_index_into(eng, jf_path, expect_k=k)
            return eng.query(lo, hi)
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish query ({what}) failed: {e}") from e


def _count_parent_on_device(parent_bam, ref_fasta, dlo, dhi, kmer_size, threads, label):
    """The counting half of _count_parent_jellyfish with the filter taken from HBM; returns the engine (the
    caller queries it and closes it: no ``parent.jf`` is written only to be read back)."""
    logger.info("%s: scanning BAM (%s): %s", label, _format_file_size(parent_bam), parent_bam)
    scan_start = time.monotonic()
    logger.info("  BAM stream -> MI355X count --if (k=%d, threads=%d, filter_kmers=%d)", kmer_size, threads, dlo.numel())
    eng = KmerEngine(kmer_size, capacity_hint=max(int(dlo.numel()), 1), device=_device())
    try:
        eng.load_filter_dev(dlo.data_ptr(), dhi.data_ptr() if dhi is not None else None, int(dlo.numel()))
        _stream_bam(eng, parent_bam, ref_fasta, threads, filtered=True)
        _merge_filter_counts(eng, None, None, dlo, dhi)        # (several ranks: one all-reduce of the per-key counts)
    except Exception:
        eng.close()
        raise
    logger.info("  %s jellyfish counting complete (%s)", label, _format_elapsed(time.monotonic() - scan_start))
    return eng


def _filter_parents_discovery(mother_bam, father_bam, ref_fasta, child_non_ref_fa, kmer_size, threads, tmpdir,
                              parent_max_count=0):
    """Module 2: mother then father (on the survivors), keep
    ``count <= parent_max_count``.  Returns (n_proband_unique, path | None).

    The key set stays in HBM from stage to stage: it is the parent's ``--if`` filter (kdf_load_filter_dev), the
    engine that counted the parent answers the ``jellyfish query`` directly (kdf_query_dev), and the survivors
    are a device-side mask.  ``after_mother.fa`` and ``proband_unique.fa`` are written as the reference writes them."""
    try:
        dev = devkeys.lookup(child_non_ref_fa, kmer_size)
        if dev is None:
            lo, hi = read_kmer_fasta_keys(child_non_ref_fa, kmer_size)
            if len(lo) == 0:
                return 0, None
            dev = devkeys.from_host(lo, hi, kmer_size > 32)
    except (KdfError, ValueError, OSError) as e:
        raise RuntimeError(f"jellyfish count (Mother) failed: {e}") from e
    dlo, dhi = dev
    devkeys.forget(child_non_ref_fa)                             # taken: the registry must not pin GBs of HBM for the life of the process
    n_input = int(dlo.numel())
    if n_input == 0:
        return 0, None
    logger.info("Filtering %d non-reference k-mers against parents…", n_input)

    def one_parent(bam, label, dlo, dhi):
        os.makedirs(os.path.join(tmpdir, label.lower()), exist_ok=True)
        try:
            eng = _count_parent_on_device(bam, ref_fasta, dlo, dhi, kmer_size, threads, label)
            try:
                keep = devkeys.query(eng, dlo, dhi, eng.device) <= parent_max_count
            finally:
                eng.close()
        except (KdfError, ValueError, OSError) as e:
            raise RuntimeError(f"jellyfish count ({label}) failed: {e}") from e
        return dlo[keep].contiguous(), (dhi[keep].contiguous() if dhi is not None else None)

    dlo, dhi = one_parent(mother_bam, "Mother", dlo, dhi)
    after_mother_fa = os.path.join(tmpdir, "after_mother.fa")
    n_surviving = int(dlo.numel())
    if dist_env.is_root():                                         # (the contract files are written once; every rank holds the same sets)
        write_kmer_fasta(after_mother_fa, *devkeys.to_host(dlo, dhi), kmer_size)
    logger.info("Mother: %d / %d non-ref k-mers found (count > %d), %d surviving",
                n_input - n_surviving, n_input, parent_max_count, n_surviving)
    if n_surviving == 0:
        return 0, None

    dlo, dhi = one_parent(father_bam, "Father", dlo, dhi)
    proband_unique_fa = os.path.join(tmpdir, "proband_unique.fa")
    n_proband = int(dlo.numel())
    if dist_env.is_root():
        write_kmer_fasta(proband_unique_fa, *devkeys.to_host(dlo, dhi), kmer_size)
        remove_with_sidecar(after_mother_fa)
    dist_env.barrier()
    logger.info("Father: %d / %d surviving k-mers found (count > %d), %d proband-unique",
                n_surviving - n_proband, n_surviving, parent_max_count, n_proband)
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
