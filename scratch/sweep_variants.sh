#!/bin/bash
# usage (GPU box): scratch/sweep_variants.sh  -- bench every scratch/variants/libkdf_*.so (kernel C thread/EPB variants)
cd $GRAFT_REPO_ROOT
cp kmer_denovo_filter_amd/libkdf.so /tmp/libkdf_keep.so
for f in scratch/variants/libkdf_*.so; do
  cp $f kmer_denovo_filter_amd/libkdf.so
  timeout -k 10 120 python bench.py --no-cpu-baseline $SWEEP_ARGS > /tmp/b.json 2>/dev/null || { echo "$f FAILED"; continue; }
  python3 - "$f" <<'PY'
import json,sys
d=json.load(open("/tmp/b.json")); r=d["roofline"]
print(sys.argv[1], d["value"], "pass", r["avg_launch_ms"], "C", r["stage_avg_ms"]["kb_bucket_kernel"], "ge3", d["config"]["kmers_ge3"], "distinct", d["config"]["distinct_per_gpu"])
PY
done
cp /tmp/libkdf_keep.so kmer_denovo_filter_amd/libkdf.so
