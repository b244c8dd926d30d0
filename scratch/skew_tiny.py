import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmer_denovo_filter_amd import KmerEngine, ReadStream
R1 = "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTGGTGTTAACCTTAGTATACTCCCTCTCCGGGCTCTGGCTCATAGGAGCAAGTCGTTGCGCTTTTAAATGTAGCCAGTGATCTTGGTTGGAACAAGGCCTACGGAAGCGCAACTCCGTCG"
R2 = "TTAACGAGCTCCTTACCGGTAGGAGTAGGAGTACACCGCAGGAAGGACTAGTCGCGGTGTGTAGAGGAACGGGAGCGCGATATGACCGCATTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTGTTTTTTTTTTTTTTT"
def run(name, reads, path=3):
    e = KmerEngine(31, capacity_hint=1 << 20); e.set_option("force_path", path)
    e.count(ReadStream.from_strings(reads))
    lo, hi, cnt = e.export_ge(0)
    z = [(int(c)) for l, c in zip(lo, cnt) if l == 0]
    d = lo[1:][lo[1:] == lo[:-1]]
    print(f"{name:34s} path {path}: distinct {len(lo):5d} key0 counts {z} dup keys {[hex(int(x)) for x in d[:4]]}", flush=True)
    e.close()
cases = {
 "R1+R2": [R1, R2], "R2+R1": [R2, R1], "R1 alone": [R1], "R2 alone": [R2], "R1+R1": [R1, R1], "R2+R2": [R2, R2],
 "T31+G.., T43": ["T" * 31 + R1[31:], "C" + "T" * 43 + "C"],
 "T31G, T43": ["T" * 31 + "G", "T" * 43],
 "T31, T43": ["T" * 31, "T" * 43],
 "T31, T32": ["T" * 31, "T" * 32],
 "A31, A43": ["A" * 31, "A" * 43],
 "T31, T31": ["T" * 31, "T" * 31],
 "T43, T43": ["T" * 43, "T" * 43],
 "T43, T44": ["T" * 43, "T" * 44],
 "T31, A31": ["T" * 31, "A" * 31],
 "CA20.., CA30..": ["CA" * 20, "CA" * 30],
}
for n, r in cases.items():
    run(n, r)
run("R1+R2", [R1, R2], 2)
