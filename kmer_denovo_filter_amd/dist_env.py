"""The process group the drop-in mirrors work in.

The reference runs one process per sample stage (``samtools fasta | jellyfish count``).  The MI355X mirrors shard a
stage over the GPUs of a node when they are started as one process per GPU -- ``torchrun --nproc-per-node N`` or any
launcher that sets ``RANK`` / ``WORLD_SIZE`` and calls ``torch.distributed.init_process_group`` ("nccl" = RCCL over
xGMI; "gloo" in tests, where the ranks share one GPU and the collectives are staged through the host).  Without an
initialised process group everything runs as the reference does: one process, one GPU.
"""
from __future__ import annotations


def world_rank():
    """(world size, rank, stage the collectives through the host?) of the initialised default process group, else (1, 0, False)."""
    try:
        import torch.distributed as dist
    except Exception:  # noqa: BLE001
        return 1, 0, False
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, False
    return dist.get_world_size(), dist.get_rank(), dist.get_backend() == "gloo"


def is_root() -> bool:
    return world_rank()[1] == 0


def barrier():
    w, _, _ = world_rank()
    if w > 1:
        import torch.distributed as dist
        dist.barrier()


def all_gather_keys(lo, hi):
    """Every rank's (lo, hi) device key tensors, concatenated in rank order on every rank (ragged sizes)."""
    w, _, host = world_rank()
    if w == 1:
        return lo, hi
    import torch
    import torch.distributed as dist
    dev = lo.device
    n = torch.tensor([lo.numel()], dtype=torch.int64, device="cpu" if host else dev)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])

    def gather(t):
        pad = torch.zeros(mx, dtype=t.dtype, device="cpu" if host else dev)
        pad[:t.numel()] = t.cpu() if host else t
        outs = [torch.empty_like(pad) for _ in range(w)]
        dist.all_gather(outs, pad)
        return torch.cat([o[:s] for o, s in zip(outs, sizes)]).to(dev)

    return gather(lo), (gather(hi) if hi is not None else None)
