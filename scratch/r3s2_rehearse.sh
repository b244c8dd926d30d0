#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2_reh; mkdir -p $O
for cfg in "weak:" "strong:--scaling strong --batches 4" "pf:--config parent_filter --scaling strong"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --no-cpu-baseline $fl > $O/$name.json 2> $O/$name.err; rc=$?
  echo "$name rc=$rc"; tail -c 1500 $O/$name.json; echo
  [ $rc -ne 0 ] && tail -20 $O/$name.err
done
