#!/bin/bash
cd $GRAFT_REPO_ROOT
KDF_EXTRA_FLAGS="-DKB_C_AGG" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err
SKEW_PATHS=1 timeout -k 10 600 python scratch/skew_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['genome'], d['wall_ms'], d['stage_ms'])"
