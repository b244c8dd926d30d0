"""repro of the fused-dump fuzz failure: seed 505, case 13 (runs the cases before it to keep the rng in step)"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import oracle as O
O.build()
import test_gpu_fuzz as F
from kmer_denovo_filter_amd import KmerEngine
orig = KmerEngine.export_ge_dev
def spy(self, min_count, d_lo, d_hi, d_cnt, cap, sorted_=False):
    st0 = {n: self.get_stat(n) for n in ("pending_passes", "flushes", "fused_dumps", "heavy_buckets", "replayed_buckets", "log2cap", "bucket_bits", "binned_passes")}
    mylo = torch.zeros(max(cap, 1), dtype=torch.int64, device="cuda:0"); myhi = torch.zeros(max(cap, 1), dtype=torch.int64, device="cuda:0"); mycnt = torch.zeros(max(cap, 1), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    n = orig(self, min_count, mylo.data_ptr(), myhi.data_ptr() if d_hi else None, mycnt.data_ptr(), cap, sorted_)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(d_lo), ctypes.c_void_p(mylo.data_ptr()), ctypes.c_size_t(cap * 8), 3)
    if d_hi: hip.hipMemcpy(ctypes.c_void_p(d_hi), ctypes.c_void_p(myhi.data_ptr()), ctypes.c_size_t(cap * 8), 3)
    hip.hipMemcpy(ctypes.c_void_p(d_cnt), ctypes.c_void_p(mycnt.data_ptr()), ctypes.c_size_t(cap * 4), 3)
    if st0["pending_passes"]:
        tl, th, tc = self.export_ge(0)
        gl = mylo[:n].cpu().numpy().view(np.uint64); gh = myhi[:n].cpu().numpy().view(np.uint64); gc = mycnt[:n].cpu().numpy().view(np.uint32)
        table = {(int(a), int(b)): int(c) for a, b, c in zip(tl, th, tc)}
        vals, cnts = np.unique(gc, return_counts=True)
        print("  dumped counts histogram", dict(zip(vals.tolist(), cnts.tolist())), "unique keys", len(set(zip(gl.tolist(), gh.tolist()))))
        bad = [(hex(int(a)), hex(int(b)), int(c), table.get((int(a), int(b)))) for a, b, c in zip(gl, gh, gc) if table.get((int(a), int(b))) != int(c)]
        print("  entries whose dumped count differs from the table's:", len(bad), bad[:8])
        print("  table: distinct", len(tl), "with count >=", min_count, int((tc >= min_count).sum()))
    st1 = {k: self.get_stat(k) for k in st0}
    print("export_ge_dev", min_count, "cap", cap, "->", n, "before", st0, "after", st1, flush=True)
    return n
KmerEngine.export_ge_dev = spy
rng = np.random.default_rng(505)
for it in range(14):
    print("case", it, flush=True)
    F.round2_case(O, rng, 1, f"seed 505 case {it}")
