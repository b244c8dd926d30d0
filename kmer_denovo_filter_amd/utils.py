"""The reference's k-mer FASTA helpers by name (``kmer_denovo_filter/utils.py`` :150-222:
``_write_kmer_fasta``, ``_load_kmers_from_fasta``, ``_estimate_fasta_sequence_count``) for callers that
hold k-mers as Python strings.  They keep the reference's arguments, results and file format
(``>{i}\\n{KMER}\\n``, i from 0) but are built on buffered whole-file operations; the engine's own stages
use the vectorised binary codec in ``kmer_fasta.py``."""
from __future__ import annotations

import itertools
import os

_WRITE_BLOCK = 1 << 16          # records per write call


def _write_kmer_fasta(kmers, filepath):
    """One record per k-mer, numbered from 0, in blocks of _WRITE_BLOCK records."""
    numbered = enumerate(kmers)
    with open(filepath, "w") as out:
        while True:
            block = list(itertools.islice(numbered, _WRITE_BLOCK))
            if not block:
                break
            out.write("".join(">%d\n%s\n" % rec for rec in block))


def _load_kmers_from_fasta(fasta_path):
    """The distinct sequence lines of the file (header lines and blank lines dropped)."""
    with open(fasta_path) as src:
        return {ln for ln in src.read().split("\n") if ln and ln[0] != ">"}


def _estimate_fasta_sequence_count(fasta_path, sample_lines=1000):
    """(number of records, whether it is an extrapolation).  The first ``sample_lines`` lines are read;
    a file that ends inside the sample is counted exactly, a longer one is scaled by
    file size / sampled bytes (at least 1)."""
    if sample_lines <= 0:
        raise ValueError("sample_lines must be > 0")
    try:
        total_bytes = os.path.getsize(fasta_path)
    except OSError:
        return 0, False
    if not total_bytes:
        return 0, False
    with open(fasta_path, "rb") as src:
        head = list(itertools.islice(src, sample_lines))
    headers = sum(1 for ln in head if ln.lstrip()[:1] == b">")
    seen = sum(map(len, head))
    if not seen or not headers:
        return 0, False
    if len(head) < sample_lines:              # the file ended inside the sample
        return headers, False
    return max(1, int(round(headers * total_bytes / seen))), True
