#!/usr/bin/env python3
"""Per-kernel HBM bytes from the two PMC passes of scripts/collect_traffic.sh.

Corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read,
so it is doubled; WRITE_SIZE is taken as is.  Our loads are 8 B per lane rather
than the calibrated 16 B, so the piece-sort kernel -- which reads and writes
exactly 8 B per entry -- is printed as a calibration point next to its known
byte count.
"""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def per_kernel(sub, counter):
    f = glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv"))
    acc, n, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, n


fetch, nf = per_kernel("fetch", "FETCH_SIZE")
write, nw = per_kernel("write", "WRITE_SIZE")
bench = json.load(open(os.path.join(out, "fetch", "bench.json")))
cfg = bench["config"]
n_entries = cfg.get("windows_rank0", cfg.get("windows_per_gpu"))
pipeline = cfg.get("count_path", "binned")
wide = cfg["k"] > 32
# prefixes of the kernels of one count pass, per pipeline
names = {
    "binned": ("kb_slabsort_kernel<", "kb_groupsum_kernel", "kb_binscan_kernel<", "kb_binfirst_kernel", "kb_piecesort_kernel<", "kb_piecesort_pipe_kernel<",
               "kb_piecesort_more_kernel<", "kb_bucket_kernel<", "kb_heavy_slice_kernel", "kb_heavy_combine_kernel", "kb_replay_kernel<"),
}.get(pipeline, ("kdf_stream_kernel<",))
rows, total = [], 0.0
for k in sorted(set(fetch) | set(write)):
    if not (k.startswith("kb_") or k.startswith("kdf_") or k.startswith("sk_")):
        continue
    rd = fetch.get(k, 0.0) * 1024 * 2          # KiB -> B, gfx950 half-count correction
    wr = write.get(k, 0.0) * 1024
    rows.append({"kernel": k, "dispatches": int(nf.get(k, 0)), "read_bytes": rd, "write_bytes": wr})
    if k.startswith(names):
        total += rd + wr
summary = {
    "tag": tag, "reads_per_gpu": cfg.get("reads_per_batch", cfg.get("reads_per_gpu")), "k": cfg["k"], "read_len": cfg["read_len"],
    "pipeline": pipeline,
    "windows": n_entries, "distinct": cfg.get("distinct_rank0_local", cfg.get("distinct_per_gpu")), "table_slots": cfg["table_slots"],
    "hbm_bytes_per_pass": total,
    "hbm_bytes_per_window": total / n_entries,
    "kernels": rows,
    "calibration": {
        "kernel": "kb_piecesort_pipe_kernel<%d>" % (2 if wide else 1), "known_read_bytes": n_entries * (16 if wide else 8),
        "known_write_bytes": n_entries * (16 if wide else 8),
        "note": "binned pipeline: known = one entry each way (plus < 1 % offset tables)",
    },
    "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half count), WRITE_SIZE KiB x 1024",
}
os.makedirs("profiles", exist_ok=True)
json.dump(summary, open(f"profiles/traffic_{tag}.json", "w"), indent=1)
if len(sys.argv) > 3 and sys.argv[3] == "latest":          # the default bench workload: what bench.py reports as roofline.traffic
    json.dump(summary, open("profiles/traffic_latest.json", "w"), indent=1)
for r in rows:
    print(f"{r['kernel'][:36]:38s} x{r['dispatches']:<3d} read {r['read_bytes']/1e9:8.3f} GB  write {r['write_bytes']/1e9:8.3f} GB")
print("known piecesort: read %.3f GB write %.3f GB" % (n_entries * (16 if wide else 8) / 1e9, n_entries * (16 if wide else 8) / 1e9))
print("pass total %.2f GB = %.1f B/window (%s pipeline, k=%d)" % (total / 1e9, total / n_entries, pipeline, cfg["k"]))
