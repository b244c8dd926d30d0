"""The test-side BAM writer and the engine's reader agree (no GPU)."""
import os

import numpy as np

from helpers import make_ref_fasta, write_bam
from test_host_feeders import unpack


def test_writer_reader_roundtrip(oracle, tmp_path):
    from kmer_denovo_filter_amd import FLAG_OFF_MODULE3, bam_reader
    ref = make_ref_fasta(str(tmp_path / "ref.fa"))
    assert len(ref) == 200
    reads = [
        {"name": "r1", "seq": ref[10:70], "pos": 10},
        {"name": "r1", "seq": ref[80:140], "pos": 80, "flag": 0x800},          # supplementary: dropped by samtools fasta
        {"name": "r2", "seq": "ACGTN" * 10, "pos": 90, "flag": 0x400},          # duplicate: dropped
        {"name": "r3", "seq": ref[100:160], "pos": 100, "flag": 0x40 | 0x1},
        {"name": "r3", "seq": ref[100:150], "pos": 100, "flag": 0x40 | 0x1},    # same QNAME + read part: collapsed
        {"name": "r3", "seq": ref[20:60], "pos": 120, "flag": 0x80 | 0x1},
        {"name": "u1", "seq": "ACGTACGTAC", "ref": -1, "pos": -1, "flag": 4},
        {"name": "big", "seq": "ACGT" * 20000, "pos": 150},                      # spans several BGZF blocks
    ]
    path = str(tmp_path / "x.bam")
    write_bam(path, [("chr1", 300)], reads)
    assert oracle.samtools_fasta_reads(path) == [ref[10:70], ref[100:160], ref[20:60], "ACGTACGTAC", "ACGT" * 20000]
    got = []
    for st in bam_reader(path, max_bases=1 << 18):
        got.extend(unpack(st))
    assert got == [ref[10:70], ref[100:160], ref[20:60], "ACGTACGTAC", "ACGT" * 20000]
    names = []
    for st in bam_reader(path, flag_off=FLAG_OFF_MODULE3, collapse=False, max_bases=1 << 18, want_meta=True):
        names.extend(zip(st.names, st.flags.tolist()))
    assert names == [("r1", 0), ("r1", 0x800), ("r3", 0x41), ("r3", 0x41), ("r3", 0x81), ("u1", 4), ("big", 0)]
