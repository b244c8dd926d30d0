#!/bin/bash
# LDS counters of the count pass (is the LDS pipe what the kernels wait for?).  --pmc only with --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lds_r03b_k63; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -i "lds" | head -40 > $out/avail.txt; cat $out/avail.txt | cut -c1-160 | head -30
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $out/$tag --output-format csv -- python3 bench.py --k 63 --steps 2 --warmup 1 --no-cpu-baseline > $out/$tag.json 2> $out/$tag.err
done
python3 - $out <<'PY' > gpurun_out/r03b_lds_counters_k63.txt
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for fn in glob.glob(f"{out}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("kb_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("== LDS counters of the count pass, per dispatch, in millions")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_LDS", 0)):
    d = {c: acc[k][c] / max(1, len(n[(k, c)])) for c in acc[k]}
    if d.get("SQ_INSTS_LDS", 0) < 1e6: continue
    print(f"  {k[:40]:40s} " + "  ".join(f"{c[3:]} {d[c] / 1e6:9.1f}" for c in sorted(d)))
PY
cat gpurun_out/r03b_lds_counters_k63.txt
