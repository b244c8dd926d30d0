#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "-DKB_C_LA=2" "-DKB_C_LA=3" "-DKB_C_LA=1"; do
echo "== $f"
KDF_EXTRA_FLAGS="$f" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err || { echo build failed; tail -3 gpurun_out/build.err; continue; }
for k in 63 31; do timeout -k 10 300 python bench.py --k $k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['k'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['stage_avg_ms']['kb_bucket_kernel'], d['config'].get('kmers_ge3'))"; done
done
