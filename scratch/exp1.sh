#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" "-DKDF_EXPORT1_COND"; do
echo "== $f"
KDF_EXTRA_FLAGS="$f" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err || { echo build failed; tail -3 gpurun_out/build.err; continue; }
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3), d['config']['kmers_ge3'])"
done
