"""Driver of the super-k-mer experiment (run on the GPU box: python prototypes/superkmer/run.py [reads])."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kmer_denovo_filter_amd import KmerEngine                      # noqa: E402
from kmer_denovo_filter_amd.synth import stream_to_ascii, synth_stream   # noqa: E402

K, M, W = 31, 12, 20
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libskproto.so"))
lib.sk_buckets.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int]
lib.sk_extract.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
MM = (1 << (2 * M)) - 1


def order(x):
    g = (x * 0x9E3779) & 0xFFFFFF
    return g ^ (g >> 11)


def model_records(reads):
    """Python model of sk_emit_kernel on ASCII reads: multiset of canonical (lo, hi|nk<<58) records."""
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    out = {}
    for r in reads:
        L = len(r)
        b = [code.get(c, -1) for c in r]
        ords = []
        for j in range(L - M + 1):
            mm = b[j:j + M]
            if min(mm) < 0:
                ords.append(None); continue
            f = 0; c = 0
            for t in range(M):
                f = (f << 2) | mm[t]; c |= (3 - mm[t]) << (2 * t)
            ords.append(order(min(f, c)))
        mins = []
        for w in range(L - K + 1):
            if min(b[w:w + K]) < 0:
                mins.append(None)
            else:
                mins.append(min(ords[w:w + W]))
        w = 0
        while w < len(mins):
            if mins[w] is None:
                w += 1; continue
            e = w + 1
            while e < len(mins) and mins[e] is not None and mins[e] == mins[w]:
                e += 1
            nk = e - w; nb = nk + K - 1
            fw = 0; rc = 0
            for t in range(nb):
                fw |= b[w + t] << (2 * t); rc |= (3 - b[w + nb - 1 - t]) << (2 * t)
            v = min(fw, rc, key=lambda x: (x >> 64, x & ((1 << 64) - 1)))
            key = (v & ((1 << 64) - 1), (v >> 64) | (nk << 58))
            out[key] = out.get(key, 0) + 1
            w = e
    return out


def extract(ds, cap, reps=3, coarse_bits=9):
    dev = ds.packed.device
    rec = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    bkt = torch.empty(cap, dtype=torch.int32, device=dev)
    tot = (ctypes.c_uint64 * 4)(); ms = (ctypes.c_float * 2)()
    torch.cuda.synchronize()
    rc = lib.sk_extract(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases, coarse_bits, rec.data_ptr(), bkt.data_ptr(), cap, tot, ms, reps)
    assert rc == 0, rc
    return rec, bkt, [int(x) for x in tot], [float(x) for x in ms]


def bucket_hash(g):
    """sk_bucket_hash of sk_proto.hip on int64 tensors."""
    return ((g * 0x9E3779B1) & 0xFFFFFFFF) ^ (g >> 9)


def balanced_assignment(rec, g, nb_bits):
    """minimizer -> bucket table with (nearly) equal DISTINCT-k-mer weight per bucket: weight of a
    minimizer = k-mers of its distinct records (an upper bound of its distinct k-mers; a product
    version would estimate it from a sample of the stream), heaviest first, dealt out in snake order."""
    key = torch.stack([rec[:, 0], rec[:, 1], g], 1)
    uniq = torch.unique(key, dim=0)
    w = torch.zeros(1 << 24, dtype=torch.int64, device=rec.device)
    w.index_add_(0, uniq[:, 2], (uniq[:, 1] >> 58) & 63)
    order = torch.argsort(w, descending=True)
    nb = 1 << nb_bits
    pos = torch.arange(order.numel(), device=rec.device)
    rnd, off = pos // nb, pos % nb
    bucket = torch.where(rnd % 2 == 0, off, nb - 1 - off)
    table = torch.empty(1 << 24, dtype=torch.int64, device=rec.device)
    table[order] = bucket
    return table


def stage2(rec, gval, windows, distinct, ge3, nb_bits=17, versions=(1, 2), balanced=False):
    """Group the records by bucket (untimed: torch sort stands in for the A1'/B' partition of 2 GB of
    records) and run the bucket kernel; check distinct / sum / count>=3 against the engine."""
    dev = rec.device
    g = gval.to(torch.int64) & 0xFFFFFF
    if balanced:
        b = balanced_assignment(rec, g, nb_bits).index_select(0, g)
    else:
        b = bucket_hash(g) >> (32 - nb_bits)
    order = torch.argsort(b)
    nk0 = int(((rec[:, 1] >> 58) & 63).sum())
    # (advanced indexing rec[order] returned garbage for the upper half at 1.3e8 rows on this torch build: gather per column)
    rec = torch.stack([rec[:, 0].contiguous().index_select(0, order), rec[:, 1].contiguous().index_select(0, order)], 1).contiguous(); b = b.index_select(0, order)
    nk1 = int(((rec[:, 1] >> 58) & 63).sum()); half = rec.shape[0] // 2
    print('nk sums before/after gather', nk0, nk1, 'second half nk sum', int(((rec[half:, 1] >> 58) & 63).sum()), 'b monotone', bool((b[1:] >= b[:-1]).all()), flush=True)
    counts = torch.bincount(b, minlength=1 << nb_bits)
    boff = torch.zeros((1 << nb_bits) + 1, dtype=torch.int64, device=dev); boff[1:] = torch.cumsum(counts, 0)
    boff32 = boff.to(torch.int32).contiguous()
    print(('balanced ' if balanced else 'hashed ') + f"records per bucket: mean {counts.float().mean().item():.0f} max {counts.max().item()} p99 {int(torch.quantile(counts.float()[:: max(1, counts.numel() // 65536)], 0.99).item())}", flush=True)
    tab_lo = torch.empty((1 << nb_bits) * 4096, dtype=torch.int64, device=dev)
    tab_cnt = torch.empty((1 << nb_bits) * 4096, dtype=torch.int32, device=dev)
    for version in versions:
        stats = torch.zeros((1 << nb_bits, 8), dtype=torch.int64, device=dev); ms = ctypes.c_float(0)
        torch.cuda.synchronize()
        rc = lib.sk_buckets(rec.data_ptr(), boff32.data_ptr(), 1 << nb_bits, tab_lo.data_ptr(), tab_cnt.data_ptr(), stats.data_ptr(),
                            ctypes.byref(ms), 3, version)
        assert rc == 0, rc
        tot = stats.sum(0).tolist(); mx = stats.max(0).values.tolist()
        ok = tot[3] == 0 and tot[0] == distinct and tot[1] == windows and tot[2] == ge3
        print(f"stage 2 v{version} ({1 << nb_bits} buckets): {ms.value:.2f} ms; distinct {tot[0]} (engine {distinct}), sum {tot[1]} (windows {windows}), "
              f">=3 {tot[2]} (engine {ge3}), dedupe overflow {tot[3]}, k-mer inserts {tot[4]} ({100.0 * tot[4] / windows:.1f}% of windows), "
              f"max distinct k-mers/bucket {mx[0]}, max distinct records/bucket {mx[5]}, max queue {mx[7]}  {'EXACT' if ok else 'differs (see README: clamp / overflow)'}", flush=True)

def main():
    dev = torch.device("cuda:0")
    # 1. small case against the Python model (slab boundaries force a few extra cuts: compare expanded k-mer totals and
    #    require that all but a handful of records match exactly)
    ds = synth_stream(3000, 150, 200_000, seed=5, device=dev); torch.cuda.synchronize()
    rec, bkt, tot, ms = extract(ds, 1 << 20, reps=1)
    n = tot[2]
    got = {}
    r = rec[:n].cpu().numpy().view(np.uint64)
    for lo, hi in r:
        got[(int(lo), int(hi))] = got.get((int(lo), int(hi)), 0) + 1
    buf, offs = stream_to_ascii(ds, ds.n_reads)
    reads = [bytes(buf[offs[i]:offs[i + 1]]) for i in range(ds.n_reads)]
    exp = model_records(reads)
    nk_got = sum((k[1] >> 58) * c for k, c in got.items()); nk_exp = sum((k[1] >> 58) * c for k, c in exp.items())
    diff = sum(abs(got.get(k, 0) - exp.get(k, 0)) for k in set(got) | set(exp))
    print(f"small: records gpu {n} model {sum(exp.values())}, k-mers gpu {nk_got} model {nk_exp} windows {tot[1]}, record multiset diff {diff}", flush=True)
    assert nk_got == nk_exp == tot[1]
    assert diff <= 8 * (ds.n_bases // 16384 + 1), diff            # only the forced cuts at slab ends may differ
    with KmerEngine(31, capacity_hint=1 << 20) as e:
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        _, distinct, windows = e.stats(); ge3 = e.count_ge(3)
    stage2(rec[:n], bkt[:n], tot[1], distinct, ge3, nb_bits=8, versions=(1, 2, 3))
    # 2. the bench workload
    reads_n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    ds = synth_stream(reads_n, 150, 100_000_000, seed=20260417, device=dev); torch.cuda.synchronize()
    rec, bkt, tot, ms = extract(ds, 1 << 28)
    with KmerEngine(31, capacity_hint=1 << 28) as e:
        e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
        _, distinct, windows = e.stats()
        ge3 = e.count_ge(3)
    print(f"bench: windows {tot[1]} (engine {windows}), records {tot[0]} = {tot[1] / tot[0]:.2f} k-mers/record, emitted {tot[2]}; "
          f"count kernel {ms[0]:.2f} ms, emit kernel {ms[1]:.2f} ms; record bytes {tot[0] * 16 / 1e9:.2f} GB vs k-mer bytes {tot[1] * 8 / 1e9:.2f} GB", flush=True)
    assert tot[1] == windows
    n = tot[2]
    stage2(rec[:n], bkt[:n], tot[1], distinct, ge3, nb_bits=18, versions=(1, 3))
    stage2(rec[:n], bkt[:n], tot[1], distinct, ge3, nb_bits=17, versions=(1, 3), balanced=True)
    stage2(rec[:n], bkt[:n], tot[1], distinct, ge3, nb_bits=18, versions=(1,), balanced=True)
    # distinct records / k-mer inserts after dedupe (what stage 2 would insert)
    r = rec[:n]
    t0 = time.time()
    key = torch.unique(r, dim=0, return_counts=False) if n < (1 << 28) else None
    if key is not None:
        nk = (key[:, 1] >> 58) & 63
        print(f"distinct records {key.shape[0]} ({100.0 * key.shape[0] / n:.1f}%), k-mer inserts after dedupe {int(nk.sum())} "
              f"({100.0 * int(nk.sum()) / tot[1]:.1f}% of windows) [{time.time() - t0:.1f}s]", flush=True)


if __name__ == "__main__":
    main()
