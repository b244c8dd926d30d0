#!/bin/bash
# long randomised parity over the code of the second half of round 3 (pipelined piece sort, big buckets, wide heavy split)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s2_fuzz
(timeout -k 10 400 python scratch/fuzz_round2.py 41 330 1 > gpurun_out/r3s2_fuzz/s41.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s41.txt) &
(timeout -k 10 400 python scratch/fuzz_round2.py 42 330 3 > gpurun_out/r3s2_fuzz/s42.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s42.txt) &
(timeout -k 10 400 python scratch/fuzz_round2.py 43 330 10 > gpurun_out/r3s2_fuzz/s43.txt 2>&1; tail -2 gpurun_out/r3s2_fuzz/s43.txt) &
wait
grep -l "Error\|assert\|Traceback" gpurun_out/r3s2_fuzz/*.txt; echo done
