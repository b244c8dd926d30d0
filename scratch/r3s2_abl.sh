#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2c; mkdir -p $O
# 2048: no kernel C (partition timed alone).  A: +256 no write-out, +1024 no rank return.  B (old kernel = +64): +256 gather from L2, +512 no write-out, +1024 no rank return
timeout -k 10 600 python scratch/ablate2.py 2048,2112,2304,3072,3328,2368,2624,3136,2880,3904 31 > $O/abl31.txt 2>&1; grep "flags\|Error" $O/abl31.txt
# C alone cannot be ablated without a partition: flags on C only (256 entries from L2, 512 no write-back)
timeout -k 10 300 python scratch/ablate2.py 0,512 31 > $O/abl31c.txt 2>&1; grep "flags\|Error" $O/abl31c.txt
