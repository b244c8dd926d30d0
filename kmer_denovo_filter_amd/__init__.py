"""kmer_denovo_filter_amd -- MI355X-native canonical k-mer count / filter / probe
engine: a drop-in for the Jellyfish/samtools subprocess path of
jlanej/kmer_denovo_filter.

Importing the package does not touch the GPU (the reference forks worker
processes, discovery/pipeline.py:788-792); the HIP library is loaded on first
use and there is no CPU fallback.
"""
from .engine import KmerEngine, hit_positions  # noqa: F401
from .reads import (  # noqa: F401
    FLAG_OFF_MODULE3,
    FLAG_OFF_SAMTOOLS_FASTA,
    ReadStream,
    bam_reader,
    fasta_reader,
    keys_to_kmers,
    kmers_to_keys,
)

__version__ = "0.1.0"
