"""Randomised parity of the ROUND-2 / ROUND-3 code paths against the oracle (GPU box): deferred flushes over streamed
batches, the pending stream of small batches, the membership sieve for count --if, owner tables (hash_shift) and the
multi-segment merge (LDS bucket merge when the
segments are in hash order, atomic fallback otherwise).  Everything is compared bit for bit."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from kmer_denovo_filter_amd import KmerEngine, ReadStream
from oracle import oracle as O
O.build()
from test_gpu_parity_basic import rand_reads

from test_gpu_fuzz import round2_case


def one_case(rng, scale, tag0):
    return round2_case(O, rng, scale, tag0)


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time(); it = 0; paths = {0: 0, 1: 0, 2: 0}
    while time.time() - t0 < budget:
        it += 1
        paths[one_case(rng, scale, f"seed {seed} it {it}")] += 1
        if it % (20 if scale == 1 else 2) == 0:
            print(f"{it} cases ok ({time.time() - t0:.0f}s) merge paths {paths}", flush=True)
    print(f"done: {it} cases, seed {seed}, scale {scale}, merge paths (0 small, 1 LDS, 2 atomic) {paths}", flush=True)
