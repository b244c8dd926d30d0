#!/bin/bash
# SQ instruction counters of the count pass (k = 31 and k = 63).  --pmc only with --kernel-trace.
# output summary -> gpurun_out/r03_sq_counters.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_r03; rm -rf $out; mkdir -p $out
: > gpurun_out/r03_sq_counters.txt
for cfg in "k31:" "k63:--k 63"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $out/$name --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > $out/$name.json 2> $out/$name.err
  python3 - $name $out <<'PY' >> gpurun_out/r03_sq_counters.txt
import csv, glob, collections, sys, json
name, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for fn in glob.glob(f"{out}/{name}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not (k.startswith("kb_") or k.startswith("kdf_export1")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
try:
    cfg = json.loads(open(f"{out}/{name}.json").read().strip().splitlines()[-1])["config"]; win = cfg["windows_rank0"]
except Exception:
    win = 0
print(f"== {name}: per dispatch, millions of wave-level events (windows per pass: {win})")
tot = collections.defaultdict(float)
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    d = {c: acc[k][c] / max(1, len(n[k])) for c in acc[k]}
    if d.get("SQ_INSTS_VALU", 0) < 1e6: continue
    if k.startswith("kb_"):
        for c in d: tot[c] += d[c]
    print(f"  {k[:44]:44s} dispatches {len(n[k]):2d}  " + "  ".join(f"{c[3:]} {d[c] / 1e6:9.1f}" for c in sorted(d)))
if win:
    print("  count pass, per window: lane-level vector instructions (VALU x 64 lanes / windows):", round(tot["SQ_INSTS_VALU"] * 64 / win, 1),
          " scalar (wave-level x 64 / windows, for scale):", round(tot["SQ_INSTS_SALU"] * 64 / win, 1))
PY
done
cat gpurun_out/r03_sq_counters.txt
