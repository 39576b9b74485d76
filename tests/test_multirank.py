"""N > 1 path on CPU: world_size-2 gloo.  Cells shard trivially, so the distributed path is (i) shard_domain,
(ii) every rank stepping its own shard with no data-path collective, (iii) one all-gather of the per-cell output
table for the writer.  The GPU is not available here, so each rank drives the CPU oracle as the stand-in engine
(test only); the gathered table must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vic_amd import abi, domain, init_state, shard
    from vic_amd.abi import C
    from oracle.pyref import OracleModel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = abi.default_options(FULL_ENERGY=1, Nband=2)
    d = domain.make_domain(23, opt, ntile=2)
    f, sf, dmy = domain.make_forcing(d, 0, 30, start_doy=75)
    sd0, si0 = init_state.initial_state(d, f[0])
    s = shard.shard_domain(d, rank, world)
    c0, c1 = s.global_cell0, s.global_cell0 + s.ncell
    m = OracleModel(s)
    m.set_state(sd0[:, s.global_hru_ids], si0[:, s.global_hru_ids])
    acc = np.zeros((2, s.ncell))
    cell = s.hru_iparams[C["HPI_CELL"]]
    cv = s.hru_dparams[C["HPD_CV"]]
    for t in range(30):
        fx, co, ce = m.step(f[t][:, :, c0:c1], sf[t][:, c0:c1], dmy[t])
        np.add.at(acc[0], cell, fx[C["FX_RUNOFF"]] * cv)
        np.add.at(acc[1], cell, fx[C["FX_BASEFLOW"]] * cv)
    b = shard.partition_cells(d.cell_hru_offset, world)
    full = shard.gather_cell_table(acc, np.diff(b))
    # the writer's table as bench.py gathers it: float32 rows of ragged shards, to the writer's rank only
    full32 = shard.gather_cell_table(acc.astype(np.float32), np.diff(b), root=0)
    assert (full32 is None) == (rank != 0)
    if rank == 0:
        q.put((full, full32, np.diff(b)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(oracle_lib):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, full32, ncell_per_rank = q.get(timeout=300)
    assert ncell_per_rank[0] != ncell_per_rank[1]          # ragged: 23 cells on two ranks
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference result
    sys.path.insert(0, ROOT)
    from vic_amd import abi, domain, init_state
    from vic_amd.abi import C
    opt = abi.default_options(FULL_ENERGY=1, Nband=2)
    d = domain.make_domain(23, opt, ntile=2)
    f, sf, dmy = domain.make_forcing(d, 0, 30, start_doy=75)
    sd0, si0 = init_state.initial_state(d, f[0])
    m = oracle_lib.OracleModel(d)
    m.set_state(sd0, si0)
    acc = np.zeros((2, d.ncell))
    cell = d.hru_iparams[C["HPI_CELL"]]
    cv = d.hru_dparams[C["HPD_CV"]]
    for t in range(30):
        fx, co, ce = m.step(f[t], sf[t], dmy[t])
        np.add.at(acc[0], cell, fx[C["FX_RUNOFF"]] * cv)
        np.add.at(acc[1], cell, fx[C["FX_BASEFLOW"]] * cv)
    assert full.shape == acc.shape
    assert np.array_equal(full, acc)
    assert full32.dtype == np.float32 and np.array_equal(full32, acc.astype(np.float32))
