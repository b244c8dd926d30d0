#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31.json 2> $O/k31.err && show $O/k31.json k31
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so; cp scratch/variants/libkdf_timing.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python scratch/phase_times.py 31 > $O/phase31.txt 2>&1; grep -v amdgpu.ids $O/phase31.txt | tail -34
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
