#!/bin/bash
# session 2 of round 3, first GPU call: parity of the big-bucket kernel C + slab-kernel variants, A/B bench, strong-8, big-table probe
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_scale.py tests/test_gpu_skew.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r['stage_avg_ms'], 'ge3', d['config']['kmers_ge3'], 'slots', d['config']['table_slots'])"; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31.json 2> $O/k31.err && show $O/k31.json main
cp kmer_denovo_filter_amd/libkdf.so /tmp/libkdf_keep.so
for v in a_s0b1 a_s1b1 a_s0b0; do
  cp scratch/variants/libkdf_$v.so kmer_denovo_filter_amd/libkdf.so
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_$v.json 2> $O/k31_$v.err && show $O/k31_$v.json $v
done
cp /tmp/libkdf_keep.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31b.json 2> $O/k31b.err && show $O/k31b.json main-again
timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > $O/strong_n1.json 2> $O/strong_n1.err && show $O/strong_n1.json strong8-big
KDF_BIG_BUCKET_LOG2CAP=64 timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > $O/strong_n1_small.json 2> $O/strong_n1_small.err && show $O/strong_n1_small.json strong8-smallbuckets
hipcc --offload-arch=gfx950 -O3 scratch/micro/ua_bench.hip -o /tmp/ua_bench 2>/dev/null && timeout -k 10 120 /tmp/ua_bench > $O/ua_bench.txt 2>&1; cat $O/ua_bench.txt
timeout -k 10 500 python scratch/bigtable_probe.py > $O/bigtable.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/bigtable.txt | tail -8
