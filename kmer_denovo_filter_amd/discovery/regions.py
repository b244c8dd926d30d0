"""N1: from Module 3's informative reads to the
discovery outputs -- interval clustering, SV annotation / linking / classes and
the BED, bedGraph, read-coverage BED and BEDPE writers.  Host post-processing of
the scan kernel's per-read hits; behaviour follows
``kmer_denovo_filter/discovery/pipeline.py`` (:1111-1144 clustering, :1156-1348
writers, :1351-1546 annotation and classes) so that the reference's committed
``giab_discovery.*`` outputs are reproduced byte for byte.
"""
from __future__ import annotations

import bisect
import logging

from ..core import bam_scanner

logger = logging.getLogger(__name__)


def _cluster_read_hits(read_hits, merge_distance):
    """Sort hits by (chrom name, start) and sweep: a hit joins the open region
    while it starts within merge_distance of the region's running end."""
    regions, region_reads, region_kmers = [], {}, {}
    if not read_hits:
        return regions, region_reads, region_kmers
    hits = sorted(read_hits, key=lambda h: (h[0], h[1]))
    cur = None
    for chrom, start, end, name, kmers, _supp in hits:
        if cur is not None and chrom == cur[0] and start <= cur[2] + merge_distance:
            cur[2] = max(cur[2], end)
            cur[3].add(name)
            cur[4].update(kmers)
            continue
        if cur is not None:
            key = (cur[0], cur[1], cur[2])
            regions.append(key); region_reads[key] = cur[3]; region_kmers[key] = cur[4]
        cur = [chrom, start, end, {name}, set(kmers)]
    key = (cur[0], cur[1], cur[2])
    regions.append(key); region_reads[key] = cur[3]; region_kmers[key] = cur[4]
    return regions, region_reads, region_kmers


def _anchor_and_cluster(child_bam, ref_fasta, proband_unique_kmers, kmer_size, merge_distance=500, threads=1,
                        min_distinct_kmers_per_read=1, proband_unique_fa=None, proband_jf=None,
                        n_proband_unique=None, tmpdir=None, memory_limit_gb=None):
    """Module 3 (reference :615-1153).  The probe structure comes from
    *proband_jf*, *proband_unique_fa* or the k-mer set, as in the reference; the
    scan runs in this process on the GPU (no fork pool: a HIP context must not be
    forked).  Returns (regions, region_reads, total_informative, region_kmers,
    unmapped_informative, read_sv_meta, kmer_coverage, read_coverage)."""
    source = proband_jf or proband_unique_fa or proband_unique_kmers
    if source is None:
        raise ValueError("no proband-unique k-mers supplied")
    bam_scanner._init_scan_worker(source, kmer_size, min_distinct_kmers_per_read)
    (read_hits, _seen, unmapped_informative, scanned, read_sv_meta, kmer_coverage,
     read_coverage) = bam_scanner.scan_bam_module3(child_bam, kmer_size, min_distinct_kmers_per_read)
    total_informative = len(read_hits) + unmapped_informative
    logger.info("Anchoring complete: %d informative reads (%d mapped, %d unmapped) from %d scanned",
                total_informative, len(read_hits), unmapped_informative, scanned)
    regions, region_reads, region_kmers = _cluster_read_hits(read_hits, merge_distance)
    logger.info("Clustered %d mapped informative reads into %d regions", len(read_hits), len(regions))
    return (regions, region_reads, total_informative, region_kmers, unmapped_informative, read_sv_meta,
            kmer_coverage, read_coverage)


def _filter_regions(regions, region_reads, region_kmers, min_supporting_reads=1, min_distinct_kmers=1):
    """Region filter of run_discovery_pipeline (reference :2375-2395)."""
    if min_supporting_reads <= 1 and min_distinct_kmers <= 1:
        return regions
    kept = []
    for key in regions:
        if (len(region_reads.get(key, ())) >= min_supporting_reads
                and len(region_kmers.get(key, ())) >= min_distinct_kmers):
            kept.append(key)
        else:
            region_reads.pop(key, None)
            region_kmers.pop(key, None)
    return kept


def _infer_sv_type(region_a, region_b):
    return "BND" if region_a[0] != region_b[0] else "INTRA"


def _annotate_and_link_from_metadata(regions, region_reads, read_sv_meta):
    """Per-region split_reads / discordant_pairs / max_clip_len / unmapped_mates and
    breakpoint links (SA targets, shared query names); reference :1351-1489."""
    by_read = {}
    for key in regions:
        for q in region_reads.get(key, ()):
            by_read.setdefault(q, set()).add(key)
    ann = {r: {"split_reads": 0, "discordant_pairs": 0, "max_clip_len": 0, "unmapped_mates": 0} for r in regions}
    if not by_read:
        return ann, []
    counted_split = set()
    for (qname, _supp), meta in read_sv_meta.items():
        for key in by_read.get(qname, ()):
            a = ann[key]
            if meta["has_sa"] and (qname, key) not in counted_split:
                a["split_reads"] += 1
                counted_split.add((qname, key))
            if meta["is_paired"]:
                if meta["mate_is_unmapped"]:
                    a["unmapped_mates"] += 1
                elif not meta["is_proper_pair"]:
                    a["discordant_pairs"] += 1
            a["max_clip_len"] = max(a["max_clip_len"], meta["max_clip"])

    by_chrom = {}
    for r in regions:
        by_chrom.setdefault(r[0], []).append(r)
    starts = {}
    for chrom, rl in by_chrom.items():
        rl.sort(key=lambda x: x[1])
        starts[chrom] = [x[1] for x in rl]
    bridges = {}
    for (qname, _supp), meta in read_sv_meta.items():
        sa = meta.get("sa_str")
        if not sa or qname not in by_read:
            continue
        for entry in sa.rstrip(";").split(";"):
            f = entry.split(",")
            if len(f) < 3:
                continue
            try:
                sa_pos = int(f[1]) - 1
            except ValueError:
                continue
            if f[0] not in starts:
                continue
            i = bisect.bisect_right(starts[f[0]], sa_pos) - 1
            if i < 0:
                continue
            target = by_chrom[f[0]][i]
            if not (target[1] <= sa_pos < target[2]):
                continue
            for src in by_read[qname]:
                if src != target:
                    bridges.setdefault(tuple(sorted([src, target])), set()).add(qname)
    for qname, rs in by_read.items():
        if len(rs) >= 2:
            rl = sorted(rs)
            for i in range(len(rl)):
                for j in range(i + 1, len(rl)):
                    bridges.setdefault((rl[i], rl[j]), set()).add(qname)
    links = [{"region_a": a, "region_b": b, "supporting_reads": bridges[(a, b)], "sv_type_hint": _infer_sv_type(a, b)}
             for a, b in sorted(bridges)]
    return ann, links


def _classify_regions(regions, region_annotations, sv_links):
    """SV / SMALL / AMBIGUOUS (reference :1517-1546); updates in place."""
    linked = set()
    for link in sv_links:
        linked.add(link["region_a"]); linked.add(link["region_b"])
    for key in regions:
        a = region_annotations.get(key, {})
        s, d, u = a.get("split_reads", 0), a.get("discordant_pairs", 0), a.get("unmapped_mates", 0)
        if s >= 2 or d >= 2 or u >= 2 or key in linked:
            a["class"] = "SV"
        elif s == 0 and d == 0 and u == 0:
            a["class"] = "SMALL"
        else:
            a["class"] = "AMBIGUOUS"
        region_annotations[key] = a


def _write_bed(regions, region_reads, region_kmers, bed_path, region_annotations=None, filters=None):
    with open(bed_path, "w") as fh:
        if filters:
            fh.write("#filters: " + " ".join(f"{k}={v}" for k, v in sorted(filters.items())) + "\n")
        fh.write("#chrom\tstart\tend\treads\tunique_kmers\tsplit_reads\tdiscordant_pairs"
                 "\tmax_clip_len\tunmapped_mates\tclass\n")
        for key in regions:
            a = (region_annotations or {}).get(key, {})
            fh.write("\t".join(str(x) for x in (
                key[0], key[1], key[2], len(region_reads.get(key, ())), len(region_kmers.get(key, ())),
                a.get("split_reads", 0), a.get("discordant_pairs", 0), a.get("max_clip_len", 0),
                a.get("unmapped_mates", 0), a.get("class", "SMALL"))) + "\n")


def _runs(sorted_items):
    """Merge consecutive positions with equal values into (start, end, value)."""
    run = None
    for pos, val in sorted_items:
        if run is not None and pos == run[1] and val == run[2]:
            run[1] = pos + 1
        else:
            if run is not None:
                yield tuple(run)
            run = [pos, pos + 1, val]
    if run is not None:
        yield tuple(run)


def _write_bedgraph(kmer_coverage, bedgraph_path, read_coverage=None, min_reads=3):
    with open(bedgraph_path, "w") as fh:
        fh.write(f"#track type=bedGraph description=\"De novo k-mer coverage (unique k-mer base "
                 f"overlaps per position, min_reads>={min_reads})\"\n")
        for chrom in sorted(kmer_coverage):
            cov = kmer_coverage[chrom]
            rc = read_coverage.get(chrom, {}) if read_coverage else None
            # a filtered-out position ends the open run even when its neighbours would merge
            segment = []
            for pos in sorted(cov):
                if rc is not None and rc.get(pos, 0) < min_reads:
                    for a, b, v in _runs(segment):
                        fh.write(f"{chrom}\t{a}\t{b}\t{v}\n")
                    segment = []
                    continue
                segment.append((pos, cov[pos]))
            for a, b, v in _runs(segment):
                fh.write(f"{chrom}\t{a}\t{b}\t{v}\n")


def _write_read_coverage_bed(kmer_coverage, read_coverage, bed_path, min_reads=3):
    with open(bed_path, "w") as fh:
        fh.write(f"#track description=\"De novo k-mer read support (min_reads>={min_reads})\"\n"
                 f"#chrom\tstart\tend\tread_count\tavg_kmers_per_read\n")
        for chrom in sorted(read_coverage):
            kc = kmer_coverage.get(chrom, {})
            items = sorted((pos, (n, round(kc.get(pos, 0) / n, 1)))
                           for pos, n in read_coverage[chrom].items() if n >= min_reads)
            for a, b, (n, avg) in _runs(items):
                fh.write(f"{chrom}\t{a}\t{b}\t{n}\t{avg}\n")


def _write_bedpe(links, bedpe_path):
    with open(bedpe_path, "w") as fh:
        fh.write("#chrom1\tstart1\tend1\tchrom2\tstart2\tend2\tsv_id\tsupporting_reads\tsv_type\n")
        for i, link in enumerate(links, 1):
            a, b = link["region_a"], link["region_b"]
            fh.write(f"{a[0]}\t{a[1]}\t{a[2]}\t{b[0]}\t{b[1]}\t{b[2]}\tSV_{i}"
                     f"\t{len(link['supporting_reads'])}\t{link['sv_type_hint']}\n")
