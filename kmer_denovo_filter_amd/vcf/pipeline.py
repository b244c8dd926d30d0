"""N3: the VCF-mode producer and annotation around the
engine's parent scan -- mirrors of ``kmer_denovo_filter/vcf/pipeline.py``:

    _parse_vcf_variants     reference :747-810   (plain / gzip VCF text, no pysam)
    _collect_child_kmers    reference :619-726   (variant-spanning child k-mers -> FASTA)
    scan_parents            reference :1571-1622 (mother + father count --if, counts ADDED)
    annotate_variants       reference :1640-1724 (DKU / DKT / DKA, MIN/AVG/MAX_PKC[_ALT])

The reference fetches reads per variant through the BAM index; this reader streams the
child BAM once instead (multi-threaded inflate + parse, ~2.8 Gbase/s) and matches every
batch against the sorted variant positions with numpy (`_records_over_positions`): only the
records that overlap a variant ever become Python objects, so the cost above the stream is
proportional to the variants, as with the index.
"""
from __future__ import annotations

import collections
import gzip
import logging
import os
import statistics

import numpy as np

from ..alignment import reads_from_batch
from ..core.jellyfish_wrappers import _scan_parent_jellyfish
from ..kmer_utils import _is_symbolic, extract_variant_spanning_kmers, read_supports_alt
from ..reads import bam_reader

logger = logging.getLogger(__name__)


def _select_alt_from_gt(alts, gt):
    """(alt, non-ref allele indices carried by the genotype); falls back to the
    first ALT when the genotype carries none."""
    idx = sorted({i for i in (gt or ()) if i is not None and i > 0 and i <= len(alts)})
    if idx:
        return alts[idx[0] - 1], idx
    return (alts[0] if alts else None), []


def _parse_vcf_variants(vcf_path, proband_id=None):
    """List of dicts chrom / pos (0-based) / ref / alts / alt / id."""
    opener = gzip.open if vcf_path.endswith(".gz") else open
    variants, samples = [], []
    with opener(vcf_path, "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            f = line.rstrip("\n").split("\t")
            if line.startswith("#"):
                samples = f[9:]
                continue
            if len(f) < 5:
                continue
            alts = tuple(a for a in f[4].split(",") if a != ".") or None
            alt = alts[0] if alts else None
            if alts and len(alts) > 1:
                if proband_id is not None and proband_id in samples and len(f) > 9:
                    fmt = f[8].split(":")
                    sval = f[9 + samples.index(proband_id)].split(":")
                    gt = None
                    if "GT" in fmt and fmt.index("GT") < len(sval):
                        gt = tuple(None if x in (".", "") else int(x)
                                   for x in sval[fmt.index("GT")].replace("|", "/").split("/"))
                    alt, _ = _select_alt_from_gt(alts, gt)
                logger.warning("Multiallelic variant %s:%s; evaluating ALT %s", f[0], f[1], alt)
            variants.append({"chrom": f[0], "pos": int(f[1]) - 1, "ref": f[3], "alts": alts, "alt": alt,
                             "id": None if f[2] == "." else f[2]})
    return variants


_REF_CONSUMING_OPS = np.zeros(16, dtype=np.int64)
_REF_CONSUMING_OPS[[0, 2, 3, 7, 8]] = 1                       # M D N = X (pysam reference_end)


def _records_over_positions(batch, refs, positions_by_chrom, eligible=None, empty_span_is_one=False):
    """Indices (ascending) of the batch's records whose reference interval [start, end) holds at least one of the
    sorted positions of its chromosome, and for each the index of the FIRST such position in that chromosome's list.
    Vectorised over the batch: reference lengths from the CIGAR arrays, one searchsorted per chromosome."""
    n = batch.n_reads
    start = np.asarray(batch.positions[:n], dtype=np.int64)
    cig = np.asarray(batch.cigar, dtype=np.uint32)
    off = np.asarray(batch.cigar_offsets[:n + 1], dtype=np.int64)
    if len(cig):
        consumed = np.concatenate(([0], np.cumsum((cig >> 4).astype(np.int64) * _REF_CONSUMING_OPS[cig & 15])))
        end = start + (consumed[off[1:]] - consumed[off[:-1]])
    else:
        end = start.copy()
    if empty_span_is_one:
        end = np.where(end <= start, start + 1, end)
    rid = np.asarray(batch.ref_ids[:n], dtype=np.int64)
    ok = np.ones(n, dtype=bool) if eligible is None else np.asarray(eligible, dtype=bool).copy()
    hit = np.zeros(n, dtype=bool)
    first = np.zeros(n, dtype=np.int64)
    for r in np.unique(rid[ok]):
        if r < 0 or r >= len(refs) or refs[r] not in positions_by_chrom:
            continue
        vp = positions_by_chrom[refs[r]]
        if len(vp) == 0:
            continue
        idx = np.flatnonzero(ok & (rid == r))
        j = np.searchsorted(vp, start[idx], side="left")
        over = (j < len(vp)) & (vp[np.minimum(j, len(vp) - 1)] < end[idx])
        hit[idx[over]] = True
        first[idx[over]] = j[over]
    keep = np.flatnonzero(hit)
    return keep, first[keep]


def _variant_key(var):
    return f"{var['chrom']}:{var['pos']}:{var['ref']}:{var['alt'] if var['alt'] is not None else '.'}"


def _collect_child_kmers(child_bam, ref_fasta, variants, kmer_size, min_baseq, min_mapq, debug_kmers, kmer_fasta,
                         flush_threshold=500_000):
    """Variant-spanning child k-mers -> ``kmer_fasta`` (``>{i}\\n{KMER}\\n``, de-duplicated
    per flush batch).  Returns (total_written, {variant key: [(read name, k-mers, supports_alt)]})."""
    by_chrom = collections.defaultdict(list)
    for v in variants:
        by_chrom[v["chrom"]].append(v)
    per_variant = {_variant_key(v): [] for v in variants}
    vpos = {c: np.unique(np.asarray([v["pos"] for v in vs], dtype=np.int64)) for c, vs in by_chrom.items()}
    rd = bam_reader(child_bam, flag_off=0, collapse=False, max_bases=1 << 24, threads=4, want_aux=True)
    refs = rd.references()
    with rd:
        for batch in rd:
            n = batch.n_reads
            eligible = ((np.asarray(batch.flags[:n]) & (0x4 | 0x100 | 0x800 | 0x400)) == 0) & (np.asarray(batch.mapq[:n]) >= min_mapq)
            keep, _ = _records_over_positions(batch, refs, vpos, eligible)
            for read in reads_from_batch(batch, refs, keep.tolist()):
                rstart, rend = read.reference_start, read.reference_end
                pairs = None
                for var in by_chrom[read.reference_name]:
                    pos = var["pos"]
                    if not (rstart <= pos < rend):
                        continue
                    if var["alt"] is not None and _is_symbolic(var["alt"]):
                        continue
                    kmers = extract_variant_spanning_kmers(read, pos, kmer_size, min_baseq, ref=var["ref"],
                                                           alt=var["alt"], seq=read.query_sequence,
                                                           quals=read.query_qualities)
                    if not kmers:
                        continue
                    if pairs is None:
                        pairs = read.get_aligned_pairs(matches_only=False)
                    supports = read_supports_alt(read, pos, var["ref"], var["alt"], min_baseq=min_baseq,
                                                 aligned_pairs=pairs, seq=read.query_sequence,
                                                 quals=read.query_qualities)
                    per_variant[_variant_key(var)].append((read.query_name, kmers, supports))
    total_written = 0
    batch_set = set()
    with open(kmer_fasta, "w") as fh:
        def flush():
            nonlocal total_written
            for km in batch_set:
                fh.write(f">{total_written}\n{km}\n")
                total_written += 1
            batch_set.clear()
        for var in variants:                       # variant order, as the reference writes them
            for _name, kmers, _s in per_variant[_variant_key(var)]:
                batch_set.update(kmers)
                if len(batch_set) >= flush_threshold:
                    flush()
        if batch_set:
            flush()
    return total_written, per_variant


def scan_parents(mother_bam, father_bam, ref_fasta, kmer_fasta, kmer_size, tmpdir, threads, total_child_kmers):
    """Step 3: both parents are scanned for the child k-mers with the engine and
    their counts are ADDED (Counter.update, reference :1592,1609)."""
    found = collections.Counter()
    for label, bam in (("mother", mother_bam), ("father", father_bam)):
        found.update(_scan_parent_jellyfish(bam, ref_fasta, kmer_fasta, kmer_size, os.path.join(tmpdir, label),
                                            threads, n_filter_kmers=total_child_kmers))
    return found


def annotate_variants(variants, variant_read_kmers, parent_found_kmers):
    """Step 4: per-variant evidence (reference :1662-1724)."""
    parent_set = set(parent_found_kmers)
    out = {}
    for var in variants:
        key = _variant_key(var)
        spanning, informative, informative_alt = set(), set(), set()
        all_kmers, alt_kmers = set(), set()
        for name, kmers, supports_alt in variant_read_kmers.get(key, []):
            spanning.add(name)
            all_kmers.update(kmers)
            if supports_alt:
                alt_kmers.update(kmers)
            if not kmers.issubset(parent_set):
                informative.add(name)
                if supports_alt:
                    informative_alt.add(name)
        dkt, dku, dka = len(spanning), len(informative), len(informative_alt)

        def pkc(ks):
            c = [parent_found_kmers[x] for x in ks if x in parent_set]
            return (max(c), round(statistics.mean(c), 2), min(c)) if c else (0, 0.0, 0)
        mx, av, mn = pkc(all_kmers)
        mxa, ava, mna = pkc(alt_kmers)
        out[key] = {"dku": dku, "dkt": dkt, "dka": dka,
                    "dku_dkt": round(dku / dkt, 4) if dkt else 0.0, "dka_dkt": round(dka / dkt, 4) if dkt else 0.0,
                    "max_pkc": mx, "avg_pkc": av, "min_pkc": mn,
                    "max_pkc_alt": mxa, "avg_pkc_alt": ava, "min_pkc_alt": mna}
    return out


def informative_reads_by_variant(variants, variant_read_kmers, parent_found_kmers):
    """{variant key: names of reads with a variant-spanning k-mer absent from both
    parents} (reference :1680-1690, the input of ``_write_informative_reads``)."""
    parent_set = set(parent_found_kmers)
    out = {}
    for var in variants:
        key = _variant_key(var)
        names = {name for name, kmers, _s in variant_read_kmers.get(key, []) if not kmers.issubset(parent_set)}
        if names:
            out[key] = names
    return out


def _write_informative_reads(child_bam, ref_fasta, informative_reads_by_variant, output_bam, threads=4):
    """Child reads carrying informative k-mers -> sorted, indexed BAM, each tagged
    ``DV:Z:<comma-joined sorted variant keys>`` (reference :1307-1357).  The reference
    fetches each variant position in sorted (chrom, pos) order and writes the first
    record it meets per read name; the same record is picked here from one pass
    over the file: smallest (region rank, file order) among the records of that
    name overlapping a variant position."""
    from ..reads import write_bam_subset
    read_to_variants = {}
    for var_key, names in informative_reads_by_variant.items():
        for name in names:
            read_to_variants.setdefault(name, set()).add(var_key)
    regions = sorted({(k.split(":")[0], int(k.split(":")[1])) for k in informative_reads_by_variant})
    rank = {r: i for i, r in enumerate(regions)}
    by_chrom = collections.defaultdict(list)
    for c, p in regions:
        by_chrom[c].append(p)
    vpos = {c: np.asarray(ps, dtype=np.int64) for c, ps in by_chrom.items()}      # (regions are sorted and unique)
    best = {}                                            # name -> (region rank, ordinal)
    rd = bam_reader(child_bam, flag_off=0, collapse=False, max_bases=1 << 24, threads=threads, want_aux=True)
    refs = rd.references()
    with rd:
        for batch in rd:
            keep, first = _records_over_positions(batch, refs, vpos, None, empty_span_is_one=True)
            for i, j in zip(keep.tolist(), first.tolist()):          # only records that overlap a variant position
                name = batch.name(i)
                if name not in read_to_variants:
                    continue
                chrom = refs[int(batch.ref_ids[i])]
                cand = (rank[(chrom, by_chrom[chrom][j])], int(batch.ordinals[i]))
                if name not in best or cand < best[name]:
                    best[name] = cand
    names = sorted(best, key=lambda n: best[n][1])
    aux = [b"DVZ" + ",".join(sorted(read_to_variants[n])).encode() + b"\0" for n in names]
    return write_bam_subset(child_bam, output_bam, [best[n][1] for n in names], aux, sort_and_index=True,
                            threads=threads)
