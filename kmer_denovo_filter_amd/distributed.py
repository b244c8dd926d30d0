"""Multi-GPU layer: one process per GPU, ``torch.distributed`` (backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" in the CPU tests).  The reference has no
distributed path at all (no NCCL/MPI call site; its only
parallelism is ``jellyfish -t`` and a per-contig process pool), so this follows
SURVEY.md section 8e rather than a reference file:

* ``ShardedFilterCount`` -- the ``count --if`` stages (parent filter
  discovery/pipeline.py:462-612, VCF Step 3 vcf/pipeline.py:1587-1609).  Reads
  are independent units: every rank holds the same filter table, counts its own
  shard of the read stream, and the per-key counts (queried in the SAME key
  order on every rank, so slot layouts need not agree) are merged with ONE
  all-reduce(sum).  No other collective touches the data path.

* ``OwnerPartitionedCount`` -- the full count stage (child counting,
  discovery/pipeline.py:69-268).  Every rank counts its read shard into a local
  table, dumps (key, count) pairs on the device, sends each pair to the rank
  that owns its key (one all-to-all; pre-aggregation bounds the traffic by the
  distinct k-mers per GPU, not by the windows) and the owner sums them.  After
  the exchange rank r holds the exact global count of every key it owns; the
  ``dump -L`` threshold is then rank-local and a scalar all-reduce gives the
  global number of survivors.  This is Jellyfish's ``merge`` (jellyfish_wrappers.py:335-366)
  done over xGMI.

Both classes are written against a tiny adapter (``TableOps``) so that the
sharding / exchange logic runs under gloo on CPU tensors in the tests, with the
oracle standing in for the table; on a GPU the adapter is ``EngineOps`` (the HIP
engine through raw device pointers).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

_U32_MAX = 0xFFFFFFFF


def _lsr(x: torch.Tensor, n: int) -> torch.Tensor:
    """Logical right shift of int64 bit patterns."""
    return (x >> n) & ((1 << (64 - n)) - 1)


def owner_of(lo: torch.Tensor, hi: Optional[torch.Tensor], world: int) -> torch.Tensor:
    """Owner rank of each key: ``((hash >> 48) * world) >> 16`` with the TABLE's hash
    (csrc/kdf_device.h ``kdf_hash``; int64 arithmetic wraps like uint64).  The top
    hash bits are the top bits of a key's home slot, so on every rank the keys of
    one owner sit in one contiguous slot range and the engine can dump them grouped
    by owner without a sort (``kdf_export_parts_dev``).  It depends on the key only
    -- not on any table's size."""
    x = lo
    if hi is not None:
        x = lo ^ ((hi << 37) | _lsr(hi, 27))
    h = (x ^ _lsr(x, 32)) * -0x604DE39AE16720DB           # KDF_MIX_MUL = 0x9FB21C651E98DF25 as int64
    return (_lsr(h, 48) * world) >> 16


class TableOps:
    """What the distributed logic needs from a table (see EngineOps)."""
    wide: bool
    device: torch.device

    def clear(self): raise NotImplementedError
    def count_stream(self, packed: torch.Tensor, invalid: torch.Tensor, n_bases: int): raise NotImplementedError
    def count_stream_filtered(self, packed: torch.Tensor, invalid: torch.Tensor, n_bases: int): raise NotImplementedError
    def export_pairs(self, min_count: int) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]: raise NotImplementedError
    def add_pairs(self, lo: torch.Tensor, hi: Optional[torch.Tensor], cnt: torch.Tensor): raise NotImplementedError
    def query(self, lo: torch.Tensor, hi: Optional[torch.Tensor]) -> torch.Tensor: raise NotImplementedError
    def count_ge(self, min_count: int) -> int: raise NotImplementedError
    def stats(self) -> Tuple[int, int, int]: raise NotImplementedError

    def add_pairs_segments(self, segments):
        """Sum the segments [(lo, hi, cnt), ...] received from the source ranks (a table that can merge them
        in one pass overrides this)."""
        for lo, hi, cnt in segments:
            self.add_pairs(lo, hi, cnt)

    def prepare_owner(self, world: int):
        """Called once on the table that will hold the keys this rank OWNS out of ``world`` ranks."""


class EngineOps(TableOps):
    """TableOps over a KmerEngine; tensors are int64 / int32 views of the
    engine's uint64 / uint32 arrays in HBM."""

    def __init__(self, engine, device: torch.device):
        self.e = engine
        self.wide = engine.wide
        self.device = device

    def clear(self):
        self.e.clear()

    def _sync(self):
        torch.cuda.current_stream(self.device).synchronize()

    def count_stream(self, packed, invalid, n_bases):
        self._sync()
        self.e.count_dev(packed.data_ptr(), invalid.data_ptr(), n_bases)

    def count_stream_filtered(self, packed, invalid, n_bases):
        self._sync()
        self.e.count_filtered_dev(packed.data_ptr(), invalid.data_ptr(), n_bases)

    def export_pairs(self, min_count):
        n = self.e.count_ge(min_count)
        lo = torch.empty(n, dtype=torch.int64, device=self.device)
        hi = torch.empty(n, dtype=torch.int64, device=self.device) if self.wide else None
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        self._sync()
        if n:
            got = self.e.export_ge_dev(min_count, lo.data_ptr(), hi.data_ptr() if hi is not None else None,
                                       cnt.data_ptr(), n)
            assert got == n
        return lo, hi, cnt

    def export_pairs_by_owner(self, world: int):
        """(lo, hi, cnt, per-owner counts) already grouped by owner rank, or None when
        the table is too small for the engine's owner-ordered dump."""
        if self.e.get_stat("log2cap") < 28 or world > 64 or self.e.get_stat("hash_shift"):
            return None
        _, distinct, _ = self.e.stats()
        lo = torch.empty(distinct, dtype=torch.int64, device=self.device)
        hi = torch.empty(distinct, dtype=torch.int64, device=self.device) if self.wide else None
        cnt = torch.empty(distinct, dtype=torch.int32, device=self.device)
        self._sync()
        n, counts = self.e.export_parts_dev(0, world, lo.data_ptr(), hi.data_ptr() if hi is not None else None,
                                            cnt.data_ptr(), distinct)
        self.e.synchronize()
        return lo[:n], (hi[:n] if hi is not None else None), cnt[:n], counts

    def export_packed_by_owner(self, world: int):
        """(send buffer uint8, per-owner pair counts, per-owner byte offsets [world + 1]): the owner-ordered dump written
        by the engine straight into the layout `OwnerPartitionedCount.exchange` sends -- no (lo, hi, cnt) temporaries, no
        packing copies.  None when the table cannot be dumped by owner (see export_pairs_by_owner)."""
        if self.e.get_stat("log2cap") < 28 or world > 64 or self.e.get_stat("hash_shift"):
            return None
        _, distinct, _ = self.e.stats()
        cap = distinct * (20 if self.wide else 12) + 8 * world + 8
        buf = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self._sync()
        n, counts, offs = self.e.export_parts_packed_dev(0, world, buf.data_ptr(), cap)
        self.e.synchronize()
        return buf[:offs[world]], counts, offs

    def add_pairs(self, lo, hi, cnt):
        if lo.numel() == 0:
            return
        lo, cnt = lo.contiguous(), cnt.contiguous()
        hi = hi.contiguous() if hi is not None else None
        self._sync()
        self.e.add_pairs_dev(lo.data_ptr(), hi.data_ptr() if hi is not None else None, cnt.data_ptr(), lo.numel())
        self.e.synchronize()

    def add_pairs_segments(self, segments):
        """All source ranks' segments in ONE engine call: when they arrive in hash order (the engine's
        owner-ordered dump) every table bucket is merged in LDS and written once (csrc/kdf_merge.h)."""
        segs = [(lo.contiguous(), hi.contiguous() if hi is not None else None, cnt.contiguous())
                for lo, hi, cnt in segments if lo.numel()]
        if not segs:
            return
        self._sync()
        self.e.add_pairs_multi_dev([(lo.data_ptr(), hi.data_ptr() if hi is not None else None, cnt.data_ptr(), lo.numel())
                                    for lo, hi, cnt in segs])
        self.e.synchronize()

    def prepare_owner(self, world: int):
        """An owner only ever sees keys whose top hash bits name it: its table drops floor(log2(world)) of them
        from the home slot (engine option ``hash_shift``), so the whole table is used and the senders' hash
        order is the table's bucket order.  (world not a power of two: the owner's keys cover
        2^floor(log2 world) / world of the table; ask for that much more capacity.)"""
        self.e.set_option("hash_shift", max(0, int(world).bit_length() - 1))

    def query(self, lo, hi):
        out = torch.zeros(lo.numel(), dtype=torch.int32, device=self.device)
        if lo.numel():
            lo = lo.contiguous()
            hi = hi.contiguous() if hi is not None else None
            self._sync()
            self.e.query_dev(lo.data_ptr(), hi.data_ptr() if hi is not None else None, lo.numel(), out.data_ptr())
            self.e.synchronize()
        return out

    def count_ge(self, min_count):
        return self.e.count_ge(min_count)

    def stats(self):
        return self.e.stats()


def _u32(t: torch.Tensor) -> torch.Tensor:
    """int32 bit patterns -> non-negative int64 values."""
    return t.to(torch.int64) & _U32_MAX


class ShardedFilterCount:
    """count --if over a read stream sharded across ranks, merged by one all-reduce."""

    def __init__(self, ops: TableOps, group=None, stage_through_host: bool = False):
        self.ops = ops
        self.group = group
        self.host = stage_through_host
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.last_reduce_dtype = None

    def merged_counts(self, keys_lo: torch.Tensor, keys_hi: Optional[torch.Tensor]) -> torch.Tensor:
        """Global count of every filter key (same key order on every rank), saturating at
        2^32-1, as int64 values.  Call after each rank counted its own shard.

        The counts travel as ONE all-reduce(sum) of 4-byte words (the `ncclUint32` reduce of
        SURVEY.md section 8e): a scalar all-reduce(max) of the largest local count first
        proves that world x max < 2^31, i.e. that the sums stay inside the SIGNED 32-bit range
        the backends reduce in (a sum in [2^31, 2^32) would be signed overflow inside gloo's /
        RCCL's `+`).  Only when that fails (counts within a factor `world` of 2^31) the sum is
        taken in 8-byte words and clamped."""
        local32 = self.ops.query(keys_lo, keys_hi)                 # int32 bit patterns of uint32 counts
        if self.world == 1 or local32.numel() == 0:
            return _u32(local32)
        if self.host:
            local32 = local32.cpu()
        mx = _u32(local32).max().reshape(1)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        if int(mx.item()) * self.world <= 0x7FFFFFFF:
            dist.all_reduce(local32, op=dist.ReduceOp.SUM, group=self.group)   # proven to stay below 2^31
            self.last_reduce_dtype = torch.int32
            out = _u32(local32)
        else:
            wide = _u32(local32)
            dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=self.group)
            self.last_reduce_dtype = torch.int64
            out = torch.clamp(wide, max=_U32_MAX)
        return out.to(keys_lo.device) if self.host else out


def _round8(n: int) -> int:
    return (n + 7) & ~7


class OwnerPartitionedCount:
    """Full count over a sharded read stream: local count, owner-partitioned
    all-to-all of (key, count) pairs, owner-side sum."""

    def __init__(self, local_ops: TableOps, group=None, device=None, owner_ops: Optional[TableOps] = None,
                 make_owner_ops=None, stage_through_host: bool = False):
        """``stage_through_host``: run the collectives on CPU copies (a gloo group over GPU
        engines: rehearsals and tests on a box without one GPU per rank; RCCL takes the device
        tensors directly)."""
        self.host = stage_through_host
        self.local = local_ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else local_ops.device
        if owner_ops is None and make_owner_ops is not None:
            owner_ops = make_owner_ops()
        if self.world > 1 and owner_ops is None:
            # summing the received pairs into the table that still holds the local partial counts would
            # count this rank's own keys twice and keep the keys it does not own
            raise ValueError("OwnerPartitionedCount: with more than one rank the owner-side table "
                             "(owner_ops or make_owner_ops) must be distinct from the local one")
        # with one rank the local table already is the global one
        self.owner = owner_ops if self.world > 1 else local_ops
        if self.world > 1 and hasattr(self.owner, "prepare_owner"):
            self.owner.prepare_owner(self.world)
        self._local_stats = (0, 0, 0)
        self.last_exchange_pairs = 0

    def local_stats(self):
        """(capacity, distinct, windows) of this rank's LOCAL table.  Reading them applies what the count calls have
        deferred: ask after the last batch, not after every batch."""
        return self.local.stats()

    def exchange(self):
        """Move every locally counted (key, count) pair to its owner rank: one count exchange and ONE
        packed all-to-all (per destination a byte segment [lo words | hi words | counts], starts aligned to 8)."""
        if self.world == 1:
            return
        packed = self.local.export_packed_by_owner(self.world) if hasattr(self.local, "export_packed_by_owner") else None
        wide = bool(getattr(self.local, "wide", False))
        esz = 20 if wide else 12
        cdev = torch.device("cpu") if self.host else self.device
        if packed is not None:                       # the engine wrote the send buffer itself, owner by owner, in hash order
            send, s_list, offs = packed
            s_bytes = [offs[p + 1] - offs[p] for p in range(self.world)]
            send_counts = torch.tensor(s_list, dtype=torch.int64)
        else:
            grouped = self.local.export_pairs_by_owner(self.world) if hasattr(self.local, "export_pairs_by_owner") else None
            if grouped is not None:                  # the engine dumps owner by owner: nothing to sort
                lo, hi, cnt, counts = grouped
                send_counts = torch.tensor(counts, dtype=torch.int64)
            else:
                lo, hi, cnt = self.local.export_pairs(0)
                own = owner_of(lo, hi, self.world)
                order = torch.argsort(own, stable=True)
                lo, cnt = lo[order], cnt[order]
                hi = hi[order] if hi is not None else None
                send_counts = torch.bincount(own, minlength=self.world).to(torch.int64).cpu()
            wide = hi is not None
            esz = 20 if wide else 12
            s_list = send_counts.tolist()
            s_bytes = [_round8(n * esz) for n in s_list]
            send = torch.empty(sum(s_bytes), dtype=torch.uint8, device=lo.device)
            off = a = 0
            for n, nb in zip(s_list, s_bytes):       # 2-3 contiguous device copies per destination
                if n:
                    send[off:off + 8 * n].view(torch.int64).copy_(lo[a:a + n])
                    o2 = off + 8 * n
                    if hi is not None:
                        send[o2:o2 + 8 * n].view(torch.int64).copy_(hi[a:a + n])
                        o2 += 8 * n
                    send[o2:o2 + 4 * n].view(torch.int32).copy_(cnt[a:a + n])
                off += nb
                a += n
        sc = send_counts.to(cdev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        r_list: List[int] = rc.tolist()
        self.last_exchange_pairs = int(sum(s_list))
        r_bytes = [_round8(n * esz) for n in r_list]
        src = send.cpu() if self.host else send
        recv = torch.empty(sum(r_bytes), dtype=torch.uint8, device=src.device)
        dist.all_to_all_single(recv, src, output_split_sizes=r_bytes, input_split_sizes=s_bytes, group=self.group)
        if self.host:
            recv = recv.to(send.device)
        off = 0
        segments = []
        for n, nb in zip(r_list, r_bytes):           # the owner sums straight from the received segments, all sources at once
            if n:
                rlo = recv[off:off + 8 * n].view(torch.int64)
                o2 = off + 8 * n
                rhi = None
                if wide:
                    rhi = recv[o2:o2 + 8 * n].view(torch.int64)
                    o2 += 8 * n
                segments.append((rlo, rhi, recv[o2:o2 + 4 * n].view(torch.int32)))
            off += nb
        if hasattr(self.owner, "add_pairs_segments"):
            self.owner.add_pairs_segments(segments)
        else:
            for seg in segments:
                self.owner.add_pairs(*seg)

    def clear(self):
        self.local.clear()
        if self.owner is not self.local:
            self.owner.clear()

    def count_local(self, packed, invalid, n_bases: int):
        """Count one more batch of this rank's read shard into its LOCAL table (no
        communication: a streamed sample is many such batches, then one merge)."""
        if isinstance(packed, int):
            raise TypeError("pass the stream tensors, not raw pointers")
        self.local.count_stream(packed, invalid, n_bases)

    def merge(self, min_count: int = 1) -> int:
        """Exchange the local (key, count) pairs to their owners; returns the global
        number of keys with count >= min_count (``dump -L``)."""
        self.exchange()
        n = torch.tensor([self.owner.count_ge(min_count)], dtype=torch.int64, device="cpu" if self.host else self.device)
        if self.world > 1:
            dist.all_reduce(n, op=dist.ReduceOp.SUM, group=self.group)
        return int(n.item())

    def count_and_merge(self, packed, invalid, n_bases: int, min_count: int = 1) -> int:
        """clear -> count the local shard -> exchange -> global number of keys
        with count >= min_count."""
        self.clear()
        self.count_local(packed, invalid, n_bases)
        return self.merge(min_count)
