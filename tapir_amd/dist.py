"""Multi-GPU layer: loci sharded round-robin over ranks, ONE all-gather of the per-locus PI tables.

Replaces `Pool(cpu_count() - 1).map(worker, params)` (bin/tapir_compute.py:159-164): loci are independent
units, so there is no data-path exchange at all until the per-locus result rows
[net PI | PI at --times | interval integrals | interval errors] are collected -- a single
`all_gather_into_tensor` of a [ceil(L/G), W] fp64 block per rank (RCCL over xGMI when the backend is
"nccl", gloo in the CPU tests).  Per-site rates stay on the rank that owns the locus (that rank writes the
locus' .rates JSON); only rank 0 needs the gathered table to write the sqlite file.

The payload is tiny (C4: 6250 x 112 doubles = 5.6 MB per rank), so the collective is latency-bound; one
process per GPU, launched by torch.distributed.run.
"""
import os

# (multi-process GPU work on this pool needs dmabuf IPC; see bench.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    if world == 1 or dist.is_initialized():
        return rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_loci(nloci, rank, world):
    """Indices of the loci rank `rank` owns: locus i -> rank i mod world (BASELINE.json config C4)."""
    return np.arange(rank, nloci, world, dtype=np.int64)


def shard_size(nloci, world):
    """Rows every rank contributes to the all-gather (the last block is padded)."""
    return (nloci + world - 1) // world


def gather_tables(local_tables, nloci, rank, world, buffers=None, always=False):
    """All-gather the per-rank [n_local, W] table blocks and undo the round-robin permutation.

    local_tables: torch tensor (CUDA for nccl, CPU for gloo) with the rows of shard_loci(nloci, rank, world)
    in that order.  Returns a [nloci, W] tensor on the same device, identical on every rank.
    buffers: optional dict the padded send block and the receive buffer are kept in between calls (a hot loop then
    allocates nothing); always: run the collective even with one rank (bench.py under a 1-rank torchrun exercises the
    code the 8-rank run executes)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return local_tables
    per = shard_size(nloci, world)
    W = local_tables.shape[1]
    if buffers is not None and "block" in buffers and buffers["block"].shape == (per, W):
        block, gathered = buffers["block"], buffers["gathered"]
    else:
        block = torch.zeros((per, W), dtype=local_tables.dtype, device=local_tables.device)
        gathered = torch.empty((world * per, W), dtype=local_tables.dtype, device=local_tables.device)
        if buffers is not None:
            buffers["block"], buffers["gathered"] = block, gathered
    block[:local_tables.shape[0]] = local_tables   # the padding rows (at most one per rank) stay zero
    dist.all_gather_into_tensor(gathered, block)
    # row r*per + j of `gathered` is locus j*world + r
    out = gathered.view(world, per, W).transpose(0, 1).reshape(per * world, W)
    return out[:nloci]
