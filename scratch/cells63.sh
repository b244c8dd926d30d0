#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, json, time
sys.path.insert(0, '.')
import torch
from kmer_denovo_filter_amd import KmerEngine
from kmer_denovo_filter_amd.synth import synth_stream
ds = synth_stream(10_000_000, 150, 100_000_000, seed=20260417, device="cuda", genome_seed=20260417); torch.cuda.synchronize()
for k in (63, 47, 33):
    ref = None
    for cells in (0, 2):
        e = KmerEngine(k, capacity_hint=1 << 28); e.set_option("binned_cells", cells)
        best = 1e9
        for it in range(4):
            e.clear(); e.synchronize(); e.profile(True)
            e.count_dev(ds.packed.data_ptr(), ds.invalid.data_ptr(), ds.n_bases); e.synchronize()
            ms, _, _ = e.profile_read(); st, _ = e.profile_stages(); e.profile(False)
            best = min(best, ms)
        s = e.stats()
        n3 = e.count_ge(3)
        if ref is None: ref = (s, n3)
        print(json.dumps({"k": k, "cells": cells, "pass_ms": round(best, 2), "stages": [round(x, 2) for x in st], "stats": s, "ge3": n3, "same": (s, n3) == ref, "cells_active": e.get_stat("binned_cells")}), flush=True)
        e.close()
PY
