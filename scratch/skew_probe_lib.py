import numpy as np


def repeat_rich(n_bases, seed):
    rng = np.random.default_rng(seed)
    alu = rng.integers(0, 4, 300)
    out, n = [], 0
    while n < n_bases:
        piece = rng.integers(0, 4, int(rng.integers(600, 2400))); out.append(piece); n += len(piece)
        copy = alu.copy(); mut = rng.random(300) < 0.10; copy[mut] = rng.integers(0, 4, int(mut.sum()))
        out.append(copy if rng.random() < 0.5 else (3 - copy)[::-1]); n += 300
        unit = [np.array([1, 0]), np.array([2, 0, 0]), np.array([0])][int(rng.integers(0, 3))]
        sat = np.tile(unit, int(rng.integers(40, 200)) // len(unit) + 1); out.append(sat); n += len(sat)
    return np.concatenate(out).astype(np.uint8)[:n_bases]

