"""Mirror of the k-mer FASTA helpers of ``kmer_denovo_filter/utils.py``
(reference :150-222): ``_write_kmer_fasta``, ``_load_kmers_from_fasta``,
``_estimate_fasta_sequence_count`` -- same names, arguments and results.  Bulk
paths inside the engine use the vectorised codec in ``kmer_fasta.py``."""
from __future__ import annotations

import os


def _write_kmer_fasta(kmers, filepath):
    """Write k-mers as ``>{i}\\n{kmer}\\n`` (reference :150-154)."""
    with open(filepath, "w") as fh:
        for i, kmer in enumerate(kmers):
            fh.write(f">{i}\n{kmer}\n")


def _load_kmers_from_fasta(fasta_path):
    """Set of the sequence lines of a k-mer FASTA (reference :157-170)."""
    kmers = set()
    with open(fasta_path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line and not line.startswith(">"):
                kmers.add(line)
    return kmers


def _estimate_fasta_sequence_count(fasta_path, sample_lines=1000):
    """(count, extrapolated) from a sampled prefix (reference :173-222)."""
    if sample_lines <= 0:
        raise ValueError("sample_lines must be > 0")
    try:
        file_size = os.path.getsize(fasta_path)
    except OSError:
        return 0, False
    if file_size == 0:
        return 0, False
    sampled_bytes = sampled_entries = lines_read = 0
    hit_eof = False
    with open(fasta_path, "rb") as fh:
        while lines_read < sample_lines:
            line = fh.readline()
            if not line:
                hit_eof = True
                break
            sampled_bytes += len(line)
            lines_read += 1
            stripped = line.strip()
            if stripped and stripped.startswith(b">"):
                sampled_entries += 1
    if sampled_bytes == 0 or sampled_entries == 0:
        return 0, False
    if hit_eof:
        return sampled_entries, False
    return max(int(round((sampled_entries / sampled_bytes) * file_size)), 1), True
