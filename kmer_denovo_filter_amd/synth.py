"""Synthetic read streams for benchmarks and tests:
uniform random genome, reads sampled at uniform starts, strand flipped with
p = 0.5, substitution errors, a fraction of bases set to N.  Built with torch
ops on whatever device is asked for (torch is plumbing here: device memory and
RNG), and packed straight into the engine's stream layout (include/kdf.h), so
the timed region of bench.py starts with the stream resident in HBM.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class DeviceStream:
    packed: torch.Tensor      # int64 words (bit pattern of the uint64 stream words), padded
    invalid: torch.Tensor     # int64 words, padded with all-ones
    n_bases: int              # stream positions (reads * (read_len + 1))
    n_reads: int
    read_len: int


def _stream_words(n_bases: int):
    tiles = (n_bases + 63) // 64
    return tiles * 2 + 4, tiles + 2


def synth_genome(genome_len: int, seed: int, device="cuda") -> torch.Tensor:
    dev = torch.device(device)
    gg = torch.Generator(device=dev)
    gg.manual_seed(seed)
    return torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=gg)


def plant_snvs(genome: torch.Tensor, rate: float, seed: int) -> torch.Tensor:
    """Copy of ``genome`` with a fraction ``rate`` of positions substituted
    (the child-private variants of the synthetic trio)."""
    g = torch.Generator(device=genome.device)
    g.manual_seed(seed)
    hit = torch.rand(genome.numel(), device=genome.device, generator=g) < rate
    alt = torch.randint(1, 4, (genome.numel(),), dtype=torch.uint8, device=genome.device, generator=g)
    return torch.where(hit, (genome + alt) & 3, genome)


def genome_stream(genome: torch.Tensor) -> DeviceStream:
    """The genome itself as one record (for the reference index)."""
    dev = genome.device
    n = genome.numel() + 1
    pw, mw = _stream_words(n)
    pad = (-n) % 64
    codes = torch.cat([genome, torch.zeros(1 + pad, dtype=torch.uint8, device=dev)])
    inv = torch.zeros(n + pad, dtype=torch.bool, device=dev)
    inv[genome.numel():] = True
    sh2 = (2 * torch.arange(32, device=dev, dtype=torch.int64))
    sh1 = torch.arange(64, device=dev, dtype=torch.int64)
    packed = torch.zeros(pw, dtype=torch.int64, device=dev)
    invalid = torch.full((mw,), -1, dtype=torch.int64, device=dev)
    step = 1 << 24
    for a in range(0, n + pad, step):
        b = min(n + pad, a + step)
        w = (codes[a:b].reshape(-1, 32).to(torch.int64) << sh2[None, :]).sum(dim=1)
        m = (inv[a:b].reshape(-1, 64).to(torch.int64) << sh1[None, :]).sum(dim=1)
        packed[a // 32: a // 32 + w.numel()] = w
        invalid[a // 64: a // 64 + m.numel()] = m
    return DeviceStream(packed, invalid, n, 1, genome.numel())


def synth_stream(n_reads: int, read_len: int = 150, genome_len: int = 100_000_000,
                 seed: int = 20260417, device="cuda", sub_rate: float = 0.005,
                 n_rate: float = 0.001, chunk_reads: int = 1 << 19,
                 genome_seed: int | None = None, genome: torch.Tensor | None = None) -> DeviceStream:
    """Returns the packed stream of ``n_reads`` synthetic reads on ``device``.

    ``genome_seed`` (default: ``seed``) fixes the genome; ranks of a multi-GPU
    run share the genome and differ in ``seed`` (their read shard).  An explicit
    ``genome`` tensor (uint8 codes) overrides both.
    """
    dev = torch.device(device)
    if genome is None:
        genome = synth_genome(genome_len, seed if genome_seed is None else genome_seed, dev)
    else:
        genome_len = genome.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    L1 = read_len + 1
    n_bases = n_reads * L1
    pw, mw = _stream_words(n_bases)
    packed = torch.zeros(pw, dtype=torch.int64, device=dev)
    invalid = torch.full((mw,), -1, dtype=torch.int64, device=dev)
    ar = torch.arange(read_len, device=dev)
    sh2 = (2 * torch.arange(32, device=dev, dtype=torch.int64))
    sh1 = torch.arange(64, device=dev, dtype=torch.int64)
    chunk_reads = max(64, (chunk_reads // 64) * 64)     # chunk * L1 stays a multiple of 64
    done = 0
    while done < n_reads:
        m = min(chunk_reads, n_reads - done)
        starts = torch.randint(0, genome_len - read_len + 1, (m,), device=dev, generator=g)
        bases = genome[starts[:, None] + ar[None, :]]                       # [m, L]
        flip = torch.rand(m, device=dev, generator=g) < 0.5
        bases = torch.where(flip[:, None], 3 - bases.flip(1), bases)
        err = torch.rand(m, read_len, device=dev, generator=g) < sub_rate
        alt = torch.randint(1, 4, (m, read_len), dtype=torch.uint8, device=dev, generator=g)
        bases = torch.where(err, (bases + alt) & 3, bases)
        isn = torch.rand(m, read_len, device=dev, generator=g) < n_rate
        codes = torch.zeros(m, L1, dtype=torch.uint8, device=dev)
        codes[:, :read_len] = torch.where(isn, torch.zeros_like(bases), bases)
        inv = torch.ones(m, L1, dtype=torch.bool, device=dev)               # separator column stays 1
        inv[:, :read_len] = isn
        flat_c = codes.reshape(-1)
        flat_i = inv.reshape(-1)
        pos0 = done * L1                                                    # multiple of 64 by construction
        n = m * L1
        pad = (-n) % 64
        if pad:
            flat_c = torch.cat([flat_c, torch.zeros(pad, dtype=torch.uint8, device=dev)])
            flat_i = torch.cat([flat_i, torch.ones(pad, dtype=torch.bool, device=dev)])
        w = (flat_c.reshape(-1, 32).to(torch.int64) << sh2[None, :]).sum(dim=1)
        mk = (flat_i.reshape(-1, 64).to(torch.int64) << sh1[None, :]).sum(dim=1)
        packed[pos0 // 32: pos0 // 32 + w.numel()] = w
        invalid[pos0 // 64: pos0 // 64 + mk.numel()] = mk
        done += m
        del starts, bases, flip, err, alt, isn, codes, inv, flat_c, flat_i, w, mk
    return DeviceStream(packed, invalid, n_bases, n_reads, read_len)


def stream_to_ascii(ds: DeviceStream, first_reads: int):
    """First ``first_reads`` reads as (uint8 ASCII buffer, int64 offsets) on the host."""
    import numpy as np
    L, L1 = ds.read_len, ds.read_len + 1
    m = min(first_reads, ds.n_reads)
    n = m * L1
    pw = ds.packed[: (n + 31) // 32].cpu().numpy().view(np.uint64)
    mw = ds.invalid[: (n + 63) // 64].cpu().numpy().view(np.uint64)
    bits = np.unpackbits(pw.view(np.uint8), bitorder="little")
    codes = (bits[0::2] | (bits[1::2] << 1))[:n]
    inv = np.unpackbits(mw.view(np.uint8), bitorder="little")[:n].astype(bool)
    chars = np.frombuffer(b"ACGT", np.uint8)[codes].copy()
    chars[inv] = ord("N")
    chars = chars.reshape(m, L1)[:, :L].copy().reshape(-1)
    offs = np.arange(m + 1, dtype=np.int64) * L
    return chars, offs
