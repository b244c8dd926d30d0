#!/bin/bash
O=gpurun_out/r3s2_merge; mkdir -p $O
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
for v in main m512 m1024; do
[ $v != main ] && cp scratch/variants/libkdf_$v.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python scratch/merge_probe.py 8 > $O/merge8_$v.txt 2>&1; echo $v; grep -v amdgpu $O/merge8_$v.txt | tail -2 | cut -c1-230
done
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
