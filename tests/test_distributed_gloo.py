"""N > 1 host logic under gloo (world_size 2, CPU tensors).  The oracle stands in
for the per-rank table (TableOps adapter), so what is tested is the sharding,
the owner-partitioned all-to-all and the all-reduce merge -- the same code the
GPU ranks run over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_reads(seed, n=400, k=21):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, 6000)
    out = []
    for _ in range(n):
        L = int(rng.integers(k, 160))
        st = int(rng.integers(0, len(genome) - L))
        r = np.frombuffer(b"ACGT", np.uint8)[genome[st:st + L]].copy()
        r[rng.random(L) < 0.01] = ord("N")
        out.append(r.tobytes().decode())
    return out


class OracleOps:
    """TableOps over the CPU oracle (tests only)."""

    def __init__(self, O, k):
        self.O, self.k, self.wide = O, k, k > 32
        self.device = torch.device("cpu")
        self.clear()

    def clear(self):
        self.t = self.O.OracleTable(self.k, 1 << 12)
        self.windows = 0
        self._dict = None

    @staticmethod
    def _to_t(a, dt):
        return torch.from_numpy(a.view(dt).copy())

    def count_stream(self, reads, _invalid, _n):          # the "stream" is a list of reads here
        self.t.count_reads(reads)
        self.windows += self.O.count_windows(reads, self.k)

    def count_stream_filtered(self, reads, _invalid, _n):
        self.t.count_reads_filtered(reads)

    def export_pairs(self, min_count):
        lo, hi, c = self.t.export_ge(min_count)
        return self._to_t(lo, np.int64), (self._to_t(hi, np.int64) if self.wide else None), self._to_t(c, np.int32)

    def add_pairs(self, lo, hi, cnt):
        # insert-or-add through the oracle: load as filter (count 0) then add counts via a dict merge
        cur = getattr(self, "_dict", None)
        if cur is None:                                   # the exchange calls add_pairs once per source rank
            cur = {}
            l0, h0, c0 = self.t.export_ge(0)
            for a, b, c in zip(l0.tolist(), h0.tolist(), c0.tolist()):
                cur[(b, a)] = c
        lo_u = lo.numpy().view(np.uint64); hi_u = hi.numpy().view(np.uint64) if hi is not None else np.zeros(len(lo_u), np.uint64)
        for a, b, c in zip(lo_u.tolist(), hi_u.tolist(), (cnt.numpy().view(np.uint32)).tolist()):
            cur[(b, a)] = min(cur.get((b, a), 0) + c, 0xFFFFFFFF)
        self._dict = cur

    def count_ge(self, min_count):
        d = getattr(self, "_dict", None)
        if d is None:
            return len(self.t.export_ge(min_count)[0])
        return sum(1 for v in d.values() if v >= min_count)

    def query(self, lo, hi):
        c = self.t.query(lo.numpy().view(np.uint64), hi.numpy().view(np.uint64) if hi is not None else None)
        return self._to_t(c, np.int32)

    def stats(self):
        return (0, len(self.t), self.windows)

    def items(self):
        return dict(self._dict)


class _FixedCounts:
    """TableOps stub: query() returns fixed uint32 bit patterns (merge arithmetic only)."""
    device = torch.device("cpu")
    wide = False

    def __init__(self, counts):
        self.c = counts

    def query(self, lo, hi):
        return self.c.clone()


def _worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as O
        from kmer_denovo_filter_amd.distributed import OwnerPartitionedCount, ShardedFilterCount, owner_of
        reads = _make_reads(11)
        shard = reads[rank::world]
        # ---- full count: local count, owner exchange, owner sum
        opc = OwnerPartitionedCount(OracleOps(O, k), owner_ops=OracleOps(O, k))
        n_ge2 = opc.count_and_merge(shard, None, 0, min_count=2)
        owned = opc.owner.items()
        lo = torch.tensor([a if a < 2**63 else a - 2**64 for (_, a) in owned.keys()], dtype=torch.int64)
        hi = torch.tensor([b if b < 2**63 else b - 2**64 for (b, _) in owned.keys()], dtype=torch.int64)
        own = owner_of(lo, hi if k > 32 else None, world) if len(owned) else torch.zeros(0, dtype=torch.int64)
        assert bool((own == rank).all()), "a rank holds a key it does not own"
        # ---- count --if: replicated filter, sharded reads, one all-reduce
        full = O.OracleTable(k).count_reads(reads)
        flo, fhi, _ = full.export_ge(0)
        sel = np.arange(len(flo)) % 3 == 0
        fops = OracleOps(O, k)
        fops.t.load_filter(flo[sel], fhi[sel])
        fops.count_stream_filtered(shard, None, 0)
        sfc = ShardedFilterCount(fops)
        tl = torch.from_numpy(flo[sel].view(np.int64).copy())
        th = torch.from_numpy(fhi[sel].view(np.int64).copy()) if k > 32 else None
        merged = sfc.merged_counts(tl, th).numpy()
        assert sfc.last_reduce_dtype == torch.int32          # the counts travelled as 4-byte words
        # counts near Jellyfish's 4-byte ceiling: the 32-bit sum would wrap, so the merge falls back to 8-byte words
        big = _FixedCounts(torch.tensor([0xFFFFFFF0 - (1 << 32), 5, 0], dtype=torch.int32))
        sb = ShardedFilterCount(big)
        mb = sb.merged_counts(torch.zeros(3, dtype=torch.int64), None)
        assert sb.last_reduce_dtype == torch.int64 and mb.tolist() == [0xFFFFFFFF, 5 * world, 0]
        # per-rank counts whose SUM lands in [2^31, 2^32): no uint32 wrap, but signed overflow in a 4-byte reduction --
        # must travel as 8-byte words too, and come out exact
        per = (0x80000000 + 1000) // world + 1
        mid = _FixedCounts(torch.tensor([per, 7, 0], dtype=torch.int32))
        sm = ShardedFilterCount(mid)
        mm = sm.merged_counts(torch.zeros(3, dtype=torch.int64), None)
        assert sm.last_reduce_dtype == torch.int64 and mm.tolist() == [per * world, 7 * world, 0] and per * world >= 2**31
        # a rank-local owner table is mandatory with more than one rank (the default used to double count)
        try:
            OwnerPartitionedCount(OracleOps(O, k))
            raise AssertionError("OwnerPartitionedCount without an owner table must refuse world > 1")
        except ValueError:
            pass
        q.put((rank, n_ge2, owned, merged.tolist(), opc.local_stats()[2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,world", [(21, 2), (41, 2), (31, 3)])
def test_owner_partitioned_count_and_sharded_filter_world2(oracle, k, world):
    """(world 3: owner ranges that are not a power-of-two split of the hash space, a read shard of a different size per rank)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = _make_reads(11)
    full = oracle.OracleTable(k).count_reads(reads)
    lo, hi, cnt = full.export_ge(0)
    exp = {(int(b), int(a)): int(c) for a, b, c in zip(lo, hi, cnt)}
    got = {}
    for _, _, owned, _, _ in res:
        assert not (set(owned) & set(got)), "two ranks own the same key"
        got.update(owned)
    assert got == exp                                            # exact global counts
    assert all(r[1] == sum(1 for v in exp.values() if v >= 2) for r in res)
    assert sum(r[4] for r in res) == oracle.count_windows(reads, k)
    # count --if merge: identical on every rank and equal to the unsharded oracle
    sel = np.arange(len(lo)) % 3 == 0
    ref = oracle.OracleTable(k).load_filter(lo[sel], hi[sel]).count_reads_filtered(reads).query(lo[sel], hi[sel])
    for r in res:
        assert r[3] == ref.tolist()


def test_owner_function_is_layout_independent():
    from kmer_denovo_filter_amd.distributed import owner_of
    g = torch.Generator().manual_seed(1)
    lo = torch.randint(-2**62, 2**62, (100000,), generator=g, dtype=torch.int64)
    for world in (1, 2, 4, 8):
        o = owner_of(lo, None, world)
        assert int(o.min()) >= 0 and int(o.max()) < world
        if world > 1:
            frac = torch.bincount(o, minlength=world).double() / len(lo)
            assert float((frac - 1 / world).abs().max()) < 0.02     # balanced
    # structured keys (poly-A neighbourhoods) still spread
    lo = torch.arange(0, 100000, dtype=torch.int64)
    frac = torch.bincount(owner_of(lo, None, 8), minlength=8).double() / len(lo)
    assert float((frac - 1 / 8).abs().max()) < 0.05


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself, before anything touches the
    GPU (it used to run one rank and report n_gpus = 1).  --launch-check makes every rank report
    and exit, so this runs on a CPU-only box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert sorted(x["rank"] for x in lines) == [0, 1] and all(x["world"] == 2 and x["n_gpus"] == 2 for x in lines)
    # a launcher that started the wrong number of ranks is an error, not a silent one-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode != 0
