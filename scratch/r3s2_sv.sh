#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_sv; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_trio_golden.py tests/test_gpu_configs.py tests/test_gpu_synthetic_scenarios.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], 'survivors', d['config'].get('survivors'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
timeout -k 10 400 python bench.py --config parent_filter --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > $O/new.json 2> $O/new.err && show $O/new.json valid-only
cp scratch/variants/libkdf_sv_old.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 400 python bench.py --config parent_filter --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > $O/old.json 2> $O/old.err && show $O/old.json all-windows
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python scratch/sieve_small.py > $O/sieve_small.txt 2>&1; grep -v amdgpu $O/sieve_small.txt | tail -8 | cut -c1-200
