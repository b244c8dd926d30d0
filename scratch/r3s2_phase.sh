#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2c; mkdir -p $O
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so; cp scratch/variants/libkdf_timing.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python scratch/phase_times.py 31 > $O/phase31.txt 2>&1; grep -v amdgpu.ids $O/phase31.txt | tail -14
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
