#!/bin/bash
# usage (here, no GPU needed): scratch/build_variant.sh NAME -DFOO=1 ...   -> scratch/variants/libkdf_NAME.so
# (kernel variants for same-box A/B runs on the GPU box: scratch/sweep_variants.sh swaps them in one after the other)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=kmer_denovo_filter_amd/csrc
mkdir -p scratch/variants /tmp/kdfvar_$name
hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-atomic-optimizer-strategy=DPP "$@" -fPIC -std=c++17 -Iinclude -I$C -c $C/kdf_engine.hip -o /tmp/kdfvar_$name/kdf_engine.o
[ -f $C/kdf_sort.o ] || hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -I$C -c $C/kdf_sort.hip -o $C/kdf_sort.o
[ -f $C/kdf_host.o ] || hipcc -O2 -x c++ -fPIC -std=c++17 -Iinclude -I$C -c $C/kdf_host.cpp -o $C/kdf_host.o
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/libkdf_$name.so /tmp/kdfvar_$name/kdf_engine.o $C/kdf_sort.o $C/kdf_host.o -lz
echo built scratch/variants/libkdf_$name.so
