#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; KDF_EXTRA_FLAGS="$1" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err || { echo build failed; tail -5 gpurun_out/build.err; return; }
  python scratch/sk_probe.py 10000000 268435456 ${2:-0} | tail -1 | cut -c1-130; }
run "-DSK_C_LA=2"
run "-DSK_C_LA=3"
run "-DSK_C_LA=4"
run "-DSK_C_LA=4 -DSK_C_WQ=64"
