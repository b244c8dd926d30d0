import sys, time, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import write_bam
from kmer_denovo_filter_amd import bam_reader
path = '/tmp/synth_1m.bam'
if not os.path.exists(path):
    rng = np.random.default_rng(1)
    genome = rng.integers(0, 4, 5_000_000)
    B = np.frombuffer(b"ACGT", np.uint8)
    reads = []
    starts = np.sort(rng.integers(0, len(genome) - 150, 1_000_000))
    for i, s in enumerate(starts):
        reads.append({"name": f"r{i}", "seq": B[genome[s:s + 150]].tobytes().decode(), "pos": int(s), "flag": 0x41 if i & 1 else 0x81})
    t = time.time(); write_bam(path, [("chr1", 5_000_000)], reads); print("wrote", os.path.getsize(path) / 1e6, "MB in", time.time() - t)
for threads in (1, 4, 8):
    t = time.time(); nb = nr = 0
    for st in bam_reader(path, max_bases=1 << 26, threads=threads):
        nb += st.n_bases; nr += st.n_reads
    dt = time.time() - t
    print(f"threads={threads}: {nr} reads, {nb/1e6:.1f} Mbases in {dt:.2f}s = {nb/dt/1e6:.1f} Mbase/s, {os.path.getsize(path)/dt/1e6:.1f} MB/s compressed", flush=True)
