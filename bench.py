#!/usr/bin/env python3
"""bench.py -- canonical k-mers counted+filtered per second on MI355X.

Workload (BASELINE.json configs[1]): synthetic
10 M x 150 bp reads at k = 31 from a 100 Mbp uniform genome (seed 20260417,
0.5 % substitutions, 0.1 % N, strand flipped with p = 0.5), packed 2-bit and
resident in HBM before the timed region.  One step = one pass of the hot path
over that batch: clear the table, count every canonical k-mer (insert mode,
`jellyfish count -C`), then `dump -L 3` MATERIALISED: the surviving (key, count)
pairs are written to a preallocated HBM buffer (discovery/pipeline.py:207-226).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config count|parent_filter]

N > 1: one rank per GPU.  When no launcher set WORLD_SIZE, bench.py starts its
own N ranks (a `torch.distributed.run` child, before anything touches the GPU)
and exits with the child's code.  Every step then does what the N = 1 step does
-- clear, count, threshold -- plus the merge of the ranks' k-mers
(kmer_denovo_filter_amd/distributed.py), all inside the timed region:
  weak    every rank counts its own batch of --reads reads (work per GPU fixed);
  strong  a fixed job of --batches x --reads reads is dealt out to the ranks
          batch by batch (N = 1 counts all of it), one merge closes the step.

--config parent_filter: BASELINE.json configs[2], the `count --if` stage
(discovery/pipeline.py:377-443): synthetic 64 Mbp trio at 30x, the child's
non-reference k-mers as the filter, one parent's reads counted per step.

Prints ONE JSON line on rank 0 (contract in the task statement), including
`roofline` for the dominant kernel and `cpu_baseline` (the oracle's threaded C
port timed on this box's host cores on a bounded sample -- a reported baseline,
not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                                          # MI355X_MICROARCH.md
METRIC = "canonical k-mers counted+filtered /sec (Gk-mer/s); % HBM roofline @ k=31"


def b_alg(k: int, L: int, probe_only: bool = False) -> float:
    """Algorithmic bytes per window: 2-bit input read once + one key-slot
    read + (count read + count write; not for a probe that misses)."""
    return L / (4 * (L - k + 1)) + (8 if k <= 32 else 16) + (0 if probe_only else 8)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["count", "parent_filter"], default="count")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batches", type=int, default=8, help="strong scaling: batches of --reads reads in the fixed job")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per batch (150 bp)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--path", choices=["auto", "direct", "binned"], default="auto",
                    help="count pipeline (auto: the engine's choice; the others force it, for profiles and A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000)
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks on ONE GPU: gloo with host-staged collectives instead of RCCL "
                         "(checks the multi-GPU job end to end on a 1-GPU box; its timing means nothing)")
    ap.add_argument("--launch-check", action="store_true",
                    help="every rank reports its RANK / WORLD_SIZE and exits before any GPU call (tests)")
    return ap.parse_args(argv)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv) -> int:
    """--gpus N > 1 without a launcher: start the N ranks as ONE child process tree
    (torch.distributed.run) BEFORE this process imports torch or touches the GPU, hand
    its output through and return its exit code (non-zero if any rank failed)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        raise SystemExit(2)
    if args.launch_check:
        print(json.dumps({"launch_check": True, "rank": rank, "world": world, "n_gpus": args.gpus}), flush=True)
        return
    if args.config == "parent_filter":
        return run_parent_filter(args, world, rank, local_rank)
    return run_count(args, world, rank, local_rank)


def _init_dist(args, world, local_rank):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse_one_gpu else dev       # where the scalar collectives live
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    return torch, dist, dev, cdev, local_rank


def _cpu_baseline_count(out, ds, args, k, distinct, n_ge3, windows):
    from kmer_denovo_filter_amd.synth import stream_to_ascii
    from oracle import oracle as O          # the checker/baseline, never the product path
    O.build()
    buf, offs = stream_to_ascii(ds, args.cpu_sample_reads)
    threads = min(os.cpu_count() or 1, 16)
    cw = O.count_windows((buf, offs), k)
    cdt = None
    for _ in range(2):                      # the first run pays the page faults of ~GBs of fresh heap
        t1 = time.perf_counter()
        cd, ct, cg = O.count_tally_mt((buf, offs), k, threads, 3)
        e = time.perf_counter() - t1
        cdt = e if cdt is None else min(cdt, e)
    assert ct == cw, (ct, cw)
    out["cpu_baseline"] = {
        "value": round(cw / cdt / 1e9, 5),
        "unit": "Gk-mer/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {len(offs) - 1} reads of the same workload ({cw} windows, {cd} distinct, {cg} with "
                  f"count >= 3; {cdt:.1f}s, faster of two runs), oracle/kdf_oracle.c kdfo_count_tally_mt: "
                  "count + `dump -L 3` tally, keys dealt to one partition per thread (CPU restatement, "
                  "not Jellyfish)",
    }
    if len(offs) - 1 == args.reads:         # the whole batch was counted on the CPU: its tally must be the GPU's
        out["cpu_baseline"]["equals_gpu_result"] = bool(cd == distinct and cg == n_ge3 and ct == windows)


def run_count(args, world, rank, local_rank):
    torch, dist, dev, cdev, local_rank = _init_dist(args, world, local_rank)
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import synth_stream

    k, L = args.k, args.read_len
    strong = args.scaling == "strong"
    # batches of this rank.  weak: one batch per rank (seed by rank).  strong: the job's
    # --batches batches are dealt out round robin; N = 1 holds all of them.
    my_batches = [b for b in range(args.batches) if b % world == rank] if strong else [rank]
    t_gen = time.time()
    streams = [synth_stream(args.reads, L, args.genome, seed=20260417 + 1000 * b, device=dev, genome_seed=20260417)
               for b in my_batches]
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] generated {len(streams)} x {args.reads} x {L} bp on device in {time.time() - t_gen:.1f}s", file=sys.stderr)
    ds = streams[0] if streams else None

    # expected distinct ~ genome + error k-mers of every batch that lands in one table; size it like `-s`
    per_batch = 1 << 28 if args.reads >= 5_000_000 else max(1 << 16, args.reads * 40)
    local_hint = per_batch if len(my_batches) <= 1 else int(per_batch * (0.4 + 0.62 * len(my_batches)))
    eng = KmerEngine(k, capacity_hint=local_hint, device=local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_option("force_path", {"auto": 0, "direct": 1, "binned": 2}[args.path])
    if os.environ.get("KDF_DEBUG_FLAGS"):                  # experiments only (A/B runs of kernel variants)
        eng.set_option("debug_flags", int(os.environ["KDF_DEBUG_FLAGS"]))
    merger = None
    if world > 1:
        from kmer_denovo_filter_amd.distributed import EngineOps, OwnerPartitionedCount
        n_job = args.batches if strong else world
        owner_hint = int(per_batch * (0.4 + 0.62 * n_job) / world) + (1 << 16)
        owner_eng = KmerEngine(k, capacity_hint=owner_hint, device=local_rank)     # keys this rank owns
        owner_eng.set_stream(torch.cuda.current_stream().cuda_stream)
        merger = OwnerPartitionedCount(EngineOps(eng, dev), dist.group.WORLD, dev,
                                       owner_ops=EngineOps(owner_eng, dev), stage_through_host=args.rehearse_one_gpu)

    # room for the materialised `dump -L 3` (N = 1: the local table is the global one)
    out_cap = max(1 << 16, int(per_batch * 0.6 * max(1, len(my_batches))))
    out_lo = torch.empty(out_cap, dtype=torch.int64, device=dev)
    out_hi = torch.empty(out_cap, dtype=torch.int64, device=dev) if k > 32 else None
    out_cnt = torch.empty(out_cap, dtype=torch.int32, device=dev)

    merge_ms = [0.0]
    merge_events = []                                    # (start, stop) CUDA events around every exchange of the timed steps

    def step():
        """clear -> count this rank's batches -> [merge over the ranks] -> dump -L 3 (materialised)."""
        if merger is None:
            eng.clear()
            for s in streams:
                eng.count_dev(s.packed.data_ptr(), s.invalid.data_ptr(), s.n_bases)
            return eng.export_ge_dev(3, out_lo.data_ptr(), out_hi.data_ptr() if out_hi is not None else None,
                                     out_cnt.data_ptr(), out_cap)
        merger.clear()
        for s in streams:
            merger.count_local(s.packed, s.invalid, s.n_bases)
        # (no device-wide synchronisation around the exchange: events on the stream bracket it, read after the timed region)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        merger.exchange()                                # every pair to its owner, summed there (the `jellyfish merge`)
        e1.record()
        merge_events.append((e0, e1))
        # the same materialised dump as at N = 1, each owner its share of the keys
        n_own = owner_eng.export_ge_dev(3, out_lo.data_ptr(), out_hi.data_ptr() if out_hi is not None else None,
                                        out_cnt.data_ptr(), out_cap)
        t = torch.tensor([n_own], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if merger is None:
        _, distinct, windows = eng.stats()
    else:
        _, distinct, windows = merger.local_stats()
    merge_events.clear()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    n_ge3 = 0
    for _ in range(args.steps):
        n_ge3 = step()
    barrier()
    dt = time.perf_counter() - t0
    merge_ms[0] = sum(a.elapsed_time(b) for a, b in merge_events)
    kernel_ms, launches, positions = eng.profile_read()
    stage_ms, stage_passes = eng.profile_stages()
    stage_names = eng.profile_stage_names()
    eng.profile(False)
    if args.steps == 0:
        _, distinct, windows = (eng.stats() if merger is None else merger.local_stats())

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        w = torch.tensor([windows], dtype=torch.int64, device=cdev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        total_windows = int(w.item())
    else:
        total_windows = windows

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = max(args.steps, 1)
    value = total_windows * args.steps / dt / 1e9
    # The count is one pipeline of kernels per pass (binned path) or one
    # kernel per chunk (direct path).  The roofline is taken over the whole pass:
    # algorithmic bytes of the pass / summed duration of its kernels, HIP events
    # on the launch stream (kdf_profile*).  rocprofv3's per-kernel averages of
    # the same command (profiles/) add up to the same number.
    n_pos = sum(s.n_bases for s in streams) or 1
    win_per_pos = windows / n_pos
    alg_bytes_per_launch = (positions / max(launches, 1)) * win_per_pos * b_alg(k, L)
    avg_ms = kernel_ms / max(launches, 1)
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    staged = stage_passes > 0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if staged and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("reads_per_gpu") == args.reads and tj.get("k") == k and tj.get("read_len") == L \
                    and tj.get("pipeline", "binned") == eng.last_count_path():
                traffic = tj.get("hbm_bytes_per_pass")
        except Exception:  # noqa: BLE001
            traffic = None
    job = (f"{args.batches} x {args.reads}" if strong else f"{args.reads}") + f" x {L} bp reads"
    out = {
        "metric": METRIC,
        "value": round(value, 4),
        "unit": "Gk-mer/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64" if k <= 32 else "u128",
        "data": "synthetic" if not args.rehearse_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU over gloo; not a measurement)",
        "config": {
            "workload": f"synthetic {job} " + ("in total (strong scaling: dealt out to the ranks batch by batch)" if strong else "per GPU")
                        + f", k={k}, count+canonicalize (insert) + dump -L 3 materialised in HBM, uniform {args.genome} bp genome, seed 20260417",
            "reads_per_batch": args.reads, "batches_this_rank": len(my_batches), "read_len": L, "k": k,
            "windows_rank0": windows, "distinct_rank0_local": distinct, "kmers_ge3": int(n_ge3),
            "table_slots": eng.stats()[0], "count_path": eng.last_count_path(),
            "multi_gpu": None if world == 1 else {
                "job": "every step: clear, count the rank's batches locally, hash-ordered dump by owner, ONE packed all-to-all of "
                       "(key,count) pairs, owner-side LDS bucket merge, materialised dump -L 3 of every owner's keys, all inside "
                       "the timed region",
                "merge_ms_per_step": round(merge_ms[0] / steps, 3), "exchanged_pairs_rank0": merger.last_exchange_pairs,
                "owner_merge": {0: None, 1: "lds_buckets", 2: "global_atomics"}[owner_eng.get_stat("last_merge_path")],
            },
        },
        "roofline": {
            "bound": "hbm",
            "kernel": (eng.last_count_path() + " count pass = " + " + ".join(stage_names)) if staged else "kdf_stream_kernel<insert>",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": round(achieved * 1e9 / HBM_PEAK, 5),
            "traffic": traffic,
            "traffic_note": "HBM bytes per pass from rocprofv3 TCC counters (profiles/traffic_latest.json, "
                            "scripts/collect_traffic.sh); null when that file does not match this workload",
            "alg_bytes_per_window": b_alg(k, L),
            "launches": launches,
            "avg_launch_ms": round(avg_ms, 4),
            "stage_avg_ms": {n: round(m / stage_passes, 4) for n, m in zip(stage_names, stage_ms)} if staged else None,
        },
    }

    if world == 1 and not args.no_cpu_baseline and len(streams) == 1:
        _cpu_baseline_count(out, ds, args, k, distinct, n_ge3, windows)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_parent_filter(args, world, rank, local_rank):
    """BASELINE.json configs[2]: `jellyfish count -C --if child_non_ref.fa` over one parent's reads
    (discovery/pipeline.py:377-443).  Almost every window misses the filter, so the roofline is the
    probe-only figure of SURVEY.md section 8d (input + one key slot = 8.3125 B per window)."""
    torch, dist, dev, cdev, local_rank = _init_dist(args, world, local_rank)
    import numpy as np
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import plant_snvs, synth_genome, synth_stream

    k, L = args.k, args.read_len
    genome_len = 64_000_000 if args.genome == 100_000_000 else args.genome
    reads = int(genome_len * 30 / L) if args.reads == 10_000_000 else args.reads
    t_gen = time.time()
    ref = synth_genome(genome_len, 20260418, dev)
    child_g = plant_snvs(ref, 0.001, 20260419)
    # filter = the child's non-reference k-mers with count >= 3 (what _subtract_reference_kmers hands on)
    child = synth_stream(reads, L, genome_len, seed=20260420, device=dev, genome=child_g)
    eng = KmerEngine(k, capacity_hint=1 << 28, device=local_rank)
    torch.cuda.synchronize()                  # the engine launches on its own stream: the generated streams must be complete
    eng.count_dev(child.packed.data_ptr(), child.invalid.data_ptr(), child.n_bases)
    n3 = eng.count_ge(3)
    cand = torch.empty(n3, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.export_ge_dev(3, cand.data_ptr(), None, None, n3)
    del child
    from kmer_denovo_filter_amd.synth import genome_stream
    gs = genome_stream(ref)
    torch.cuda.synchronize()
    eng.clear()
    eng.count_dev(gs.packed.data_ptr(), gs.invalid.data_ptr(), gs.n_bases)
    in_ref = torch.empty(n3, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.query_dev(cand.data_ptr(), None, n3, in_ref.data_ptr())
    eng.synchronize()
    filt = cand[in_ref == 0].contiguous()
    n_filter = int(filt.numel())
    del cand, in_ref, gs
    # the parent's read shard of this rank (reads are independent units: SURVEY.md section 8e)
    # --scaling strong: one parent's reads are a fixed job dealt out to the ranks; weak: every rank its own `reads` reads
    per_rank = reads // world if args.scaling == "strong" else reads
    parent = synth_stream(per_rank, L, genome_len, seed=20260421 + 1000 * rank, device=dev, genome=ref)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] synthetic trio built in {time.time() - t_gen:.1f}s: filter {n_filter} keys, "
              f"{per_rank} parent reads per rank", file=sys.stderr)
    feng = KmerEngine(k, capacity_hint=max(n_filter, 1 << 10), device=local_rank)
    feng.load_filter_dev(filt.data_ptr(), None, n_filter)
    eng.close()
    counts = torch.zeros(n_filter, dtype=torch.int32, device=dev)
    sharded = None
    if world > 1:
        from kmer_denovo_filter_amd.distributed import EngineOps, ShardedFilterCount
        sharded = ShardedFilterCount(EngineOps(feng, dev), dist.group.WORLD, stage_through_host=args.rehearse_one_gpu)

    merge_s = [0.0]

    def step():
        """`count --if` over the parent shard, per-key counts in filter order [merged over the ranks],
        then the `<= parent_max_count` threshold (discovery/pipeline.py:515-538)."""
        feng.reset_counts()
        feng.count_filtered_dev(parent.packed.data_ptr(), parent.invalid.data_ptr(), parent.n_bases)
        if sharded is None:
            feng.query_dev(filt.data_ptr(), None, n_filter, counts.data_ptr())
            feng.synchronize()
            c = counts
        else:
            feng.synchronize()
            tm = time.perf_counter()
            c = sharded.merged_counts(filt, None)            # ONE all-reduce of the per-key counts
            torch.cuda.synchronize()
            merge_s[0] += time.perf_counter() - tm
        return int((c == 0).sum().item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    merge_s[0] = 0.0
    barrier()
    feng.profile(True)
    t0 = time.perf_counter()
    survivors = 0
    for _ in range(args.steps):
        survivors = step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches, positions = feng.profile_read()
    feng.profile(False)
    _, _, windows = feng.stats()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        w = torch.tensor([windows], dtype=torch.int64, device=cdev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        total_windows = int(w.item())
    else:
        total_windows = windows
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    steps = max(args.steps, 1)
    balg = b_alg(k, L, probe_only=True)
    avg_ms = kernel_ms / max(launches, 1)
    achieved = windows * balg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {
        "metric": METRIC,
        "value": round(total_windows * args.steps / dt / 1e9, 4),
        "unit": "Gk-mer/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u64" if k <= 32 else "u128",
        "data": "synthetic" if not args.rehearse_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU over gloo; not a measurement)",
        "config": {
            "workload": f"parent_filter (BASELINE configs[2] substitute, SURVEY 8d item 3): synthetic {genome_len} bp genome, "
                        f"{per_rank} x {L} bp parent reads per rank ({'a fixed job of ' + str(reads) + ' sharded over the ranks' if args.scaling == 'strong' else 'weak scaling'}), "
                        f"k={k}, count --if against the child's "
                        f"{n_filter} non-reference k-mers (0.1 % planted SNVs, count >= 3), per-key counts + <= 0 threshold",
            "filter_keys": n_filter, "parent_reads": reads, "windows_rank0": windows, "survivors": survivors,
            "count_path": feng.last_count_path(),
            "multi_gpu": None if world == 1 else {"job": "every step: count --if over the rank's shard, ONE all-reduce(sum) of the per-key counts, threshold",
                                                  "merge_ms_per_step": round(merge_s[0] / steps * 1e3, 3)},
        },
        "roofline": {
            "bound": "hbm", "kernel": feng.last_count_path() + " count --if pass",
            "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(achieved * 1e9 / HBM_PEAK, 5), "traffic": None,
            "alg_bytes_per_window": balg, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
            "model": "probe-only: 2-bit input read once + one 8 B key slot per window (windows that miss write nothing)",
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        from kmer_denovo_filter_amd.synth import stream_to_ascii
        from oracle import oracle as O
        O.build()
        sample = min(per_rank, 2_000_000)
        buf, offs = stream_to_ascii(parent, sample)
        keys = filt.cpu().numpy().view(np.uint64)
        t1 = time.perf_counter()
        ot = O.OracleTable(k, 1 << 12).load_filter(keys, np.zeros_like(keys))
        ot.count_reads_filtered((buf, offs))
        cdt = time.perf_counter() - t1
        cw = O.count_windows((buf, offs), k)
        # the GPU on the SAME sample (the first `sample` reads, re-packed as a stream of their own): per-key counts must be equal
        from kmer_denovo_filter_amd import ReadStream
        sub = ReadStream.from_ascii(buf, offs)
        with KmerEngine(k, capacity_hint=max(n_filter, 1 << 10), device=local_rank) as ge:
            ge.load_filter_dev(filt.data_ptr(), None, n_filter)
            ge.count_filtered(sub)
            gsample = ge.query(keys, None)
        sample_equal = bool(np.array_equal(gsample, ot.query(keys, np.zeros_like(keys))))
        out["cpu_baseline"] = {
            "value": round(cw / cdt / 1e9, 5), "unit": "Gk-mer/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} parent reads ({cw} windows, {cdt:.1f}s incl. loading the {n_filter}-key filter), "
                      "oracle/kdf_oracle.c count --if, one thread (CPU restatement, not Jellyfish)",
        }
        out["cpu_baseline"]["equals_gpu_result"] = sample_equal
        out["cpu_baseline"]["equals_gpu_result_on"] = f"the sample ({sample} reads): per-key counts of all {n_filter} filter k-mers"
        if sample == per_rank:
            oc = ot.query(keys, np.zeros_like(keys))
            out["cpu_baseline"]["equals_gpu_result"] = bool(sample_equal and int((oc == 0).sum()) == survivors)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
