#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "-DKB_C_HEAVY=65536u -DKB_HV_MAX=64u" "-DKB_C_HEAVY=32768u -DKB_HV_MAX=128u" "-DKB_C_HEAVY=20000u -DKB_HV_MAX=256u"; do
echo "== $f"
KDF_EXTRA_FLAGS="$f" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err || { echo build failed; tail -5 gpurun_out/build.err; continue; }
SKEW_PATHS=1 timeout -k 10 600 python scratch/skew_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['genome'], d['wall_ms'], d['stage_ms'], d['distinct'], d.get('heavy_buckets'))"
done
