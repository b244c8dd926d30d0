#!/bin/bash
set -o pipefail
ulimit -c 0
mkdir -p gpurun_out/r3c
timeout -k 10 300 python bench.py --k 63 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/k63.json 2> gpurun_out/r3c/k63.err; python3 -c "import json; d=json.load(open('gpurun_out/r3c/k63.json')); print('k63', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'])"
timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/strong_n1.json 2> gpurun_out/r3c/strong_n1.err
rc=$?; echo "strong rc=$rc"; [ $rc -ne 0 ] && { tail -5 gpurun_out/r3c/strong_n1.err; exit 1; }
python3 -c "import json; d=json.load(open('gpurun_out/r3c/strong_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'], d['config']['kmers_ge3'], d['config']['table_slots'])"
