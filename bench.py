#!/usr/bin/env python3
"""bench.py -- canonical k-mers counted+filtered per second on MI355X.

Workload (BASELINE.json configs[1]): synthetic
10 M x 150 bp reads at k = 31 from a 100 Mbp uniform genome (seed 20260417,
0.5 % substitutions, 0.1 % N, strand flipped with p = 0.5), packed 2-bit and
resident in HBM before the timed region.  One step = one pass of the hot path
over that batch: clear the table, count every canonical k-mer (insert mode,
`jellyfish count -C`), then the `dump -L 3` threshold filter as a table scan.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns a
read shard of the same size (weak scaling) and the k-mers are merged by an
owner-partitioned exchange over RCCL (kmer_denovo_filter_amd/distributed.py).

Prints ONE JSON line on rank 0 (contract in the task statement), including
`roofline` for the dominant kernel and `cpu_baseline` (the oracle's threaded C
port timed on this box's host cores on a bounded sample -- a reported baseline,
not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_K31_L150 = 150 / (4 * (150 - 31 + 1)) + 8 + 8      # 16.3125 B per window
HBM_PEAK = 8.0e12                                          # MI355X_MICROARCH.md


def b_alg(k: int, L: int) -> float:
    return L / (4 * (L - k + 1)) + (8 if k <= 32 else 16) + 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (150 bp)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000)
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks on ONE GPU: gloo with host-staged collectives instead of RCCL "
                         "(checks the multi-GPU job end to end on a 1-GPU box; its timing means nothing)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from kmer_denovo_filter_amd import KmerEngine
    from kmer_denovo_filter_amd.synth import stream_to_ascii, synth_stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    k, L = args.k, args.read_len
    t_gen = time.time()
    ds = synth_stream(args.reads, L, args.genome, seed=20260417 + 1000 * rank, device=dev,
                      genome_seed=20260417)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] generated {args.reads} x {L} bp on device in {time.time() - t_gen:.1f}s", file=sys.stderr)

    # expected distinct ~ genome + error k-mers; size the table like `-s`
    cap_hint = 1 << 28 if args.reads >= 5_000_000 else max(1 << 16, args.reads * 40)
    eng = KmerEngine(k, capacity_hint=cap_hint, device=local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    merger = None
    if world > 1:
        from kmer_denovo_filter_amd.distributed import EngineOps, OwnerPartitionedCount
        owner_eng = KmerEngine(k, capacity_hint=cap_hint, device=local_rank)     # keys this rank owns
        owner_eng.set_stream(torch.cuda.current_stream().cuda_stream)
        merger = OwnerPartitionedCount(EngineOps(eng, dev), dist.group.WORLD, dev,
                                       owner_ops=EngineOps(owner_eng, dev), stage_through_host=args.rehearse_one_gpu)

    # N = 1: a step is clear -> count the batch -> `dump -L 3` threshold (BASELINE config 2).
    # N > 1: the streamed-sample job of SURVEY.md section 8d item 4 / 8e: every rank counts
    # K batches of ITS read shard into its local table (one step each, no communication), then
    # ONE owner-partitioned exchange + owner-side sum + global threshold closes the job --
    # all inside the timed region; its share of the time is reported as config.merge_ms.
    def step():
        if merger is None:
            eng.clear()
            eng.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases)
            return eng.count_ge(3)
        merger.count_local(ds.packed, ds.invalid, ds.n_bases)
        return 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    merge_ms = 0.0
    if merger is None:
        for _ in range(args.warmup):
            step()
        _, distinct, windows = eng.stats()
    else:
        merger.clear()
        for _ in range(max(args.warmup, 1)):       # one full mini-job: warms RCCL too
            step()
        merger.merge(3)
        _, distinct, windows = merger.local_stats()
        windows //= max(args.warmup, 1)
        merger.clear()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_ge3 = step()
    if merger is not None:
        torch.cuda.synchronize()
        tm = time.perf_counter()
        n_ge3 = merger.merge(3)
        torch.cuda.synchronize()
        merge_ms = (time.perf_counter() - tm) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches, positions = eng.profile_read()
    stage_ms, stage_passes = eng.profile_stages()
    binned_passes = eng.get_stat("binned_passes")
    eng.profile(False)

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

    value = total_windows * args.steps / dt / 1e9
    # The count is one pipeline of four kernels per pass (binned path) or one
    # kernel per chunk (direct path).  The roofline is taken over the whole pass:
    # algorithmic bytes of the pass / summed duration of its kernels, HIP events
    # on the launch stream (kdf_profile*).  rocprofv3's per-kernel averages of
    # the same command (profiles/) add up to the same number.
    win_per_pos = windows / ds.n_bases
    alg_bytes_per_launch = (positions / max(launches, 1)) * win_per_pos * b_alg(k, L)
    avg_ms = kernel_ms / max(launches, 1)
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    binned = stage_passes > 0
    stage_names = ["kb_hist1_kernel(+scans)", "kb_scatter1_kernel", "kb_finesort_kernel", "kb_bucket_kernel"]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if binned and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("reads_per_gpu") == args.reads and tj.get("k") == k and tj.get("read_len") == L:
                traffic = tj.get("hbm_bytes_per_pass")
        except Exception:  # noqa: BLE001
            traffic = None
    out = {
        "metric": "canonical k-mers counted+filtered /sec (Gk-mer/s); % HBM roofline @ k=31",
        "value": round(value, 4),
        "unit": "Gk-mer/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64" if k <= 32 else "u128",
        "data": "synthetic" if not args.rehearse_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU over gloo; not a measurement)",
        "config": {
            "workload": f"synthetic {args.reads} x {L} bp reads per GPU, k={k}, count+canonicalize (insert) "
                        f"+ dump -L 3 threshold, uniform {args.genome} bp genome, seed 20260417",
            "reads_per_gpu": args.reads, "read_len": L, "k": k,
            "windows_per_gpu": windows, "distinct_per_gpu": distinct, "kmers_ge3": int(n_ge3),
            "table_slots": eng.stats()[0],
            "multi_gpu": None if world == 1 else {
                "job": f"{args.steps} local count steps per rank (the rank's synthetic batch each time, no communication), then ONE owner-partitioned "
                       "all-to-all of (key,count) pairs + owner-side sum + global dump -L 3, all inside the timed region",
                "merge_ms": round(merge_ms, 3), "exchanged_pairs_rank0": merger.last_exchange_pairs,
            },
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("binned count pass = " + " + ".join(stage_names)) if binned else "kdf_stream_kernel<insert>",
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
            "stage_avg_ms": {n: round(m / stage_passes, 4) for n, m in zip(stage_names, stage_ms)} if binned else None,
        },
    }

    if world == 1 and not args.no_cpu_baseline:
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
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
