#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; KDF_EXTRA_FLAGS="$1" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>gpurun_out/build.err || { echo build failed; tail -5 gpurun_out/build.err; return; }
  python scratch/sk_probe.py 10000000 268435456 ${2:-0} | tail -1 | cut -c1-330; }
run "-mllvm -amdgpu-atomic-optimizer-strategy=None"
run "-mllvm -amdgpu-atomic-optimizer-strategy=DPP"
run "-mllvm -amdgpu-atomic-optimizer-strategy=None -DSK_BUCKET_BITS=12 -DSK_C_THREADS=512 -DSK_C_RC=512 -DSK_C_IC=3072 -DSK_C_WQ=64"
echo "== old binned path with strategy None"
KDF_EXTRA_FLAGS="-mllvm -amdgpu-atomic-optimizer-strategy=None" python -m kmer_denovo_filter_amd.build --force > /dev/null 2>&1
python - <<'PY'
import sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda", genome_seed=20260417)
torch.cuda.synchronize()
e = KmerEngine(31, capacity_hint=1 << 28); e.set_option("force_path", 2)
for it in range(3):
    e.clear(); e.profile(True)
    e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
    ms, n = e.profile_stages(); e.profile(False)
print("binned stages", [round(x, 3) for x in ms], e.stats())
PY
