"""Multi-GPU layout of the path: cells shard trivially (no cell ever reads another cell's data).

One process per GPU.  `shard_domain` cuts a contiguous block of the lat-major cell list for a rank (balanced by HRU
count, which is what the work is proportional to); every table of include/vicgpu.h is sliced accordingly.  The only
exchange is `gather_cell_table`: the per-cell output table of every rank is all-gathered (RCCL over xGMI when the
process group is "nccl", gloo in the CPU tests) into the global (row, cell) slab the NetCDF-layout writer consumes
(WriteOutputNetCDF.c:387-455 gathers aggdata from every cell the same way).
"""
import copy

import numpy as np

from .abi import C


def partition_cells(cell_hru_offset, world):
    """Boundaries [world+1] of contiguous cell blocks with near-equal HRU counts."""
    off = np.asarray(cell_hru_offset, dtype=np.int64)
    ncell = len(off) - 1
    nhru = int(off[-1])
    bounds = [0]
    for r in range(1, world):
        target = nhru * r / world
        c = int(np.searchsorted(off, target, side="left"))
        c = min(max(c, bounds[-1]), ncell)
        bounds.append(c)
    bounds.append(ncell)
    return np.asarray(bounds, dtype=np.int64)


def shard_domain(dom, rank, world):
    """Sub-domain of `dom` for `rank`: cells [b[rank], b[rank+1]) and their HRUs, renumbered slot-major."""
    b = partition_cells(dom.cell_hru_offset, world)
    c0, c1 = int(b[rank]), int(b[rank + 1])
    nc = c1 - c0
    sub = copy.copy(dom)
    sub.opt = dom.opt
    sub.ncell = nc
    sub.cell_params = np.ascontiguousarray(dom.cell_params[:, c0:c1])
    sub.init_moist = np.ascontiguousarray(dom.init_moist[:, c0:c1])
    sub.elevation = dom.elevation[c0:c1]
    sub.cell_offset_T = dom.cell_offset_T[c0:c1]
    cell = dom.hru_iparams[C["HPI_CELL"]]
    # keep the parent's HRU order restricted to the shard: with slot-major numbering this stays slot-major
    keep = np.nonzero((cell >= c0) & (cell < c1))[0]
    new_id = -np.ones(dom.nhru, dtype=np.int64)
    new_id[keep] = np.arange(len(keep))
    sub.nhru = len(keep)
    hpi = dom.hru_iparams[:, keep].copy()
    hpi[C["HPI_CELL"]] -= c0
    sub.hru_iparams = np.ascontiguousarray(hpi)
    sub.hru_dparams = np.ascontiguousarray(dom.hru_dparams[:, keep])
    off = dom.cell_hru_offset
    lst = dom.cell_hru_list[off[c0]:off[c1]]
    sub.cell_hru_list = np.ascontiguousarray(new_id[lst].astype(np.int32))
    sub.cell_hru_offset = np.ascontiguousarray((off[c0:c1 + 1] - off[c0]).astype(np.int32))
    sub.global_cell0 = c0
    sub.global_hru_ids = keep
    return sub


def gather_cell_table(local, ncell_per_rank, group=None, device=None, root=None):
    """Gather a per-cell table [nrow][ncell_local] from every rank into [nrow][sum ncell].

    `ncell_per_rank` lists every rank's cell count (ragged shards are padded for the collective).  root = None: all-gather, every
    rank gets the table; root = r: gather to the writer's rank only (the others return None and never hold the whole table).
    Works on the "nccl" backend (= RCCL on ROCm) with device tensors and on "gloo" with CPU tensors.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nrow = local.shape[0]
    nmax = int(max(ncell_per_rank))
    src = torch.as_tensor(np.ascontiguousarray(local))          # keeps the table's dtype (float32 writer tables, float64 accumulators)
    t = torch.zeros((nrow, nmax), dtype=src.dtype, device=device)
    t[:, :local.shape[1]] = src.to(t.device)
    if root is None:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
    else:
        out = [torch.empty_like(t) for _ in range(world)] if rank == root else None
        dist.gather(t, out, dst=root, group=group)
        if rank != root:
            return None
    parts = [o[:, :int(n)].cpu().numpy() for o, n in zip(out, ncell_per_rank)]
    return np.concatenate(parts, axis=1)
