#!/bin/bash
# long randomised parity over the code of the second half of round 3 (pipelined piece sort, big buckets, wide heavy split)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s2_fuzz
(timeout -k 10 400 python scratch/fuzz_round2.py 31 330 1 > gpurun_out/r3s2_fuzz/s31.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s31.txt) &
(timeout -k 10 400 python scratch/fuzz_round2.py 32 330 6 > gpurun_out/r3s2_fuzz/s32.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s32.txt) &
(timeout -k 10 400 python scratch/fuzz_round2.py 33 330 20 > gpurun_out/r3s2_fuzz/s33.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s33.txt) &
wait
grep -l "Error\|assert\|Traceback" gpurun_out/r3s2_fuzz/*.txt; echo done
