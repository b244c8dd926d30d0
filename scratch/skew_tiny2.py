import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmer_denovo_filter_amd import KmerEngine, ReadStream
R1 = "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTGGTGTTAACCTTAGTATACTCCCTCTCCGGGCTCTGGCTCATAGGAGCAAGTCGTTGCGCTTTTAAATGTAGCCAGTGATCTTGGTTGGAACAAGGCCTACGGAAGCGCAACTCCGTCG"
rng = np.random.default_rng(3)
def both(reads):
    out = []
    for path in (1, 3):
        e = KmerEngine(31, capacity_hint=1 << 20); e.set_option("force_path", path)
        e.count(ReadStream.from_strings(reads))
        out.append((e.export_ge(0), e))
    (dl, dh, dc), de = out[0]; (sl, sh, sc), se = out[1]
    q = se.query(dl, None)
    bad = np.flatnonzero(q != dc)
    dup = sl[1:][sl[1:] == sl[:-1]]
    de.close(); se.close()
    return len(dl), len(sl), [(hex(int(dl[i])), int(dc[i]), int(q[i])) for i in bad[:4]], [hex(int(x)) for x in dup[:3]]
for pad in list(range(31, 64)) + [100, 150]:
    filler = "".join("ACGT"[x] for x in rng.integers(0, 4, pad))
    print(pad, both([filler, R1[:62]]), flush=True)
print("R1[:62] x2", both([R1[:62], R1[:62]]))
print("R1[:51] x2", both([R1[:51], R1[:51]]))
print("T31+G20 x2", both(["T" * 31 + "G" * 20] * 2))
print("T12+rand x2", both(["T" * 12 + R1[31:80]] * 2))
