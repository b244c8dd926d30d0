#!/bin/bash
# the fuzz tests, then a long randomised run (three processes) over the round-2/3 paths incl. the fused dump leg
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3s2_fuzz2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "fuzz tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
(timeout -k 10 460 python scratch/fuzz_round2.py ${S1:-41} 400 1 > $O/s_a.txt 2>&1; tail -1 $O/s_a.txt) &
(timeout -k 10 460 python scratch/fuzz_round2.py ${S2:-42} 400 6 > $O/s_b.txt 2>&1; tail -1 $O/s_b.txt) &
(timeout -k 10 460 python scratch/fuzz_round2.py ${S3:-43} 400 20 > $O/s_c.txt 2>&1; tail -1 $O/s_c.txt) &
wait
grep -l "Error\|assert\|Traceback" $O/*.txt; echo done
