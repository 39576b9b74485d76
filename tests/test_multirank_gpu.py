"""N > 1 path with the HIP library as the engine (GPU box): two ranks, each stepping ITS shard of one glacier / frozen-soil
domain (the cfg4 shape: 5 x 5 HRUs, glacier top band, 10 nodes, put_data on the device) through libvicgpu.so on cuda:0, the
writer's float table gathered to rank 0 (gloo here: one GPU cannot host two RCCL ranks; the collective itself is covered on
"nccl" by bench.py at N > 1).  The gathered table must equal the table of ONE process stepping the whole domain, bit for bit:
cells never interact, so a shard boundary must not be visible anywhere -- HRU renumbering, forcing slices, state slices,
put_data's area weights and hruList order included.  tests/test_multirank.py is the same check with the CPU oracle as the
stand-in engine for boxes without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCELL, NSTEP, RATIO = int(os.environ.get("VIC_TEST_NCELL", 37)), 24, 24      # (the override is for the host-emulated library)
OUT = ["OUT_RUNOFF", "OUT_BASEFLOW", "OUT_SWE", "OUT_SOIL_MOIST", "OUT_EVAP", "OUT_GLAC_MBAL"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    sys.path.insert(0, ROOT)
    import bench
    from vic_amd import domain, init_state
    from vic_amd.abi import C
    cfg = bench.config("cfg4")
    d = domain.make_domain(NCELL, cfg["opt"], ntile=cfg["ntile"], glacier_top_band=True)
    f, sf, dmy = domain.make_forcing(d, 0, NSTEP, start_doy=cfg["start_doy"])
    sd0, si0 = init_state.initial_state(d, f[0])
    isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
    sd0[C["SD_GLAC_CUM_MASS_BALANCE"], isg] = 0.0
    return d, f, sf, dmy, sd0, si0


def _run(dom, f, sf, dmy, sd, si):
    from vic_amd.api import Model
    m = Model(dom, device=0)
    m.set_state(sd, si)
    m.put_data_config(RATIO)
    m.put_data_init()
    m.push_forcing(f, sf, dmy)
    m.dist_prec(0, NSTEP, sync=True)
    assert m.get_cell_errors().sum() == 0
    out = m.get_outputs(OUT, reset=True)
    m.close()
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    d, f, sf, dmy, sd0, si0 = _setup()
    from vic_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = shard.shard_domain(d, rank, world)
    c0, c1 = s.global_cell0, s.global_cell0 + s.ncell
    out = _run(s, np.ascontiguousarray(f[:, :, :, c0:c1]), np.ascontiguousarray(sf[:, :, c0:c1]), dmy,
               np.ascontiguousarray(sd0[:, s.global_hru_ids]), np.ascontiguousarray(si0[:, s.global_hru_ids]))
    b = shard.partition_cells(d.cell_hru_offset, world)
    full = shard.gather_cell_table(out, np.diff(b), root=0)
    if rank == 0:
        q.put((full, np.diff(b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_of_the_hip_library_match_one_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, per_rank = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert per_rank[0] != per_rank[1] and per_rank.sum() == NCELL          # ragged shards
    d, f, sf, dmy, sd0, si0 = _setup()
    one = _run(d, f, sf, dmy, sd0, si0)
    assert full.dtype == np.float32 and full.shape == one.shape
    assert np.array_equal(full, one, equal_nan=True)
    assert np.isfinite(one[:5]).all() and np.abs(one[2]).max() > 0            # SWE present: the winter glacier domain did something
