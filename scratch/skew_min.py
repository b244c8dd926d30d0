"""find a minimal read set on which the super-k-mer path stores a key twice"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from kmer_denovo_filter_amd import KmerEngine, ReadStream
from kmer_denovo_filter_amd.synth import synth_stream, stream_to_ascii
from skew_probe_lib import repeat_rich
g = torch.from_numpy(repeat_rich(2_000_000, 7)).cuda()
ds = synth_stream(200_000, 150, seed=20260417, device="cuda", genome=g); torch.cuda.synchronize()
buf, offs = stream_to_ascii(ds, ds.n_reads)
buf = buf.tobytes()
reads = [buf[int(offs[i]):int(offs[i + 1])].decode() for i in range(ds.n_reads)]
print("reads", len(reads), "len0", len(reads[0]), flush=True)

def dups(rs, hint=1 << 20):
    e = KmerEngine(31, capacity_hint=hint); e.set_option("force_path", 3)
    e.count(ReadStream.from_strings(rs))
    lo, hi, cnt = e.export_ge(0)
    st = (e.get_stat("sk_spills"), e.get_stat("sk_failed_buckets"))
    e.close()
    d = lo[1:][lo[1:] == lo[:-1]]
    return len(d), (hex(int(d[0])) if len(d) else None), st

print("all:", dups(reads, 1 << 23), flush=True)
pa = [r for r in reads if "A" * 31 in r or "T" * 31 in r]
print("polyA reads:", len(pa), dups(pa), flush=True)
cur = pa
# bisection: keep a half that still shows duplicates
while len(cur) > 1:
    h = len(cur) // 2
    a, b = cur[:h], cur[h:]
    da, db = dups(a), dups(b)
    if da[0]: cur = a
    elif db[0]: cur = b
    else:
        print("needs both halves at", len(cur), da, db, flush=True); break
print("minimal set size", len(cur), dups(cur), flush=True)
for r in cur[:6]: print(r)
