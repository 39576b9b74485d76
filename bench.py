#!/usr/bin/env python3
"""Benchmark of the VIC hot path (dist_prec -> full_energy -> surface_fluxes) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg4|cfg2]

A "step" is one model time step of every cell of the rank's shard (all HRUs, all snow sub-steps).
Metric: cell-timesteps/s, whole job (sum over ranks).  Weak scaling: every rank owns a full copy of the
per-GPU workload (cells shard trivially, no data-path collective; BASELINE.json cfg4 = 8 x 125k cells).

Workloads (BASELINE.json configs, SURVEY.md 8(d) synthetic inputs):
  cfg3  100k cells, FULL_ENERGY + FROZEN_SOIL (10 thermal nodes, explicit, "fixed" node-parameter semantics),
        5 snow bands x 5 veg tiles = 25 HRUs/cell, hourly  -- the config the metric is quoted on (default)
  cfg4  one GPU's share (125k cells) of the 1M-cell glacier config: cfg3 with veg slot 0 of the top band a glacier HRU
  cfg2  10k cells, FULL_ENERGY (QUICK_FLUX), 1 band x 3 veg tiles, hourly

Forcing for all W+K steps is generated on the host and is resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def config(name):
    from vic_amd import abi
    if name == "cfg3":
        opt = abi.default_options(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=0)
        return dict(opt=opt, ncell=100000, ntile=5, start_doy=60,
                    workload="cfg3: 100k cells, FULL_ENERGY+FROZEN_SOIL (Nnode=10, explicit, fixed), 5 bands x 5 veg tiles, hourly")
    if name == "cfg4":       # one GPU's share of BASELINE.json configs[3] (1M cells on 8 GPUs): cfg3 + a glacier HRU in the top band
        opt = abi.default_options(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=0)
        return dict(opt=opt, ncell=125000, ntile=5, start_doy=60, glacier=True,
                    workload="cfg4 share: 125k cells (1M / 8 GPUs), FULL_ENERGY+FROZEN_SOIL (Nnode=10, explicit, fixed), 5 bands x 5 veg tiles, "
                             "veg slot 0 of the top band = glacier (solve_glacier / surface_fluxes_glac), hourly")
    if name == "cfg2":
        opt = abi.default_options(FULL_ENERGY=1)
        return dict(opt=opt, ncell=10000, ntile=3, start_doy=60,
                    workload="cfg2: 10k cells, FULL_ENERGY (QUICK_FLUX), 1 band x 3 veg tiles, hourly")
    raise SystemExit("unknown config " + name)


def b_alg(opt, hru_per_cell, n_outvar=8, out_step_ratio=24):
    """Algorithmic bytes per cell-step, SURVEY.md 8(d):
    8*NVAR_F + P_cell + sum_hru(2*S_hru + 64) + B_out with S_hru = 8*(30+Nn), P_cell = 8*(77+8*Nn+5*Nband+110)."""
    Nn, Nb = opt.Nnode, opt.Nband
    return 8 * 10 + 8 * (77 + 8 * Nn + 5 * Nb + 110) + hru_per_cell * (2 * 8 * (30 + Nn) + 64) + 4.0 * n_outvar / out_step_ratio


def cpu_baseline(cfg, target_seconds=15.0):
    """The CPU checker timed on this box's host cores on a bounded sample of the same workload.  Prefers the real
    reference build (oracle/_ref, kind "reference"); falls back to the C restatement (kind "port")."""
    from vic_amd import domain, init_state
    from oracle import pyref
    import copy
    opt = cfg["opt"]
    variant = "plain" if not opt.FROZEN_SOIL else ("compat" if opt.frozen_compat else "fixed")
    kind = "reference" if pyref.have_ref(variant) else "port"

    def run(ncell_s, nsteps):
        d = domain.make_domain(ncell_s, copy.copy(opt), ntile=cfg["ntile"], glacier_top_band=cfg.get("glacier", False))
        f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=cfg["start_doy"])
        sd0, si0 = init_state.initial_state(d, f[0])
        m = pyref.RefModel(d, variant) if kind == "reference" else pyref.OracleModel(d)
        nthreads = int(getattr(m.lib, m.prefix + "max_threads")())
        m.set_state(sd0, si0)
        secs = m.run(f, sf, dmy, nthreads)
        m.close()
        return secs, nthreads

    # calibrate on a sample big enough to give every core work, then size the timed sample for ~target_seconds
    ncell_s, nsteps = 2048, 4
    secs, nthreads = run(ncell_s, nsteps)
    rate = ncell_s * nsteps / secs
    nsteps = 12
    ncell_s = int(min(cfg["ncell"], max(2048, target_seconds * rate / nsteps)))
    secs, nthreads = run(ncell_s, nsteps)
    rate = ncell_s * nsteps / secs
    return {"value": rate, "unit": "cell-timesteps/s", "cores": nthreads, "kind": kind,
            "sample": "%d cells x %d steps of the same workload (%.1f s, OpenMP over cells, %s)" % (
                ncell_s, nsteps, secs, "reference build oracle/_ref/libvicref_%s.so" % variant if kind == "reference" else "oracle/libvicoracle.so")}


def measured_traffic(config_name, ncell):
    """HBM bytes per step from the committed PMC passes of this workload (profiles/r01_traffic.json, written by
    tools/round_summary.py from separate FETCH_SIZE / WRITE_SIZE passes); None when no measurement matches."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    if t.get("config") != config_name or t.get("cells_per_gpu") != ncell:
        return None
    return t.get("hbm_bytes_per_step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--ncell", type=int, default=0, help="override cells per GPU (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from vic_amd import domain, init_state
    from vic_amd.api import Model
    from vic_amd.abi import C

    cfg = config(args.config)
    opt = cfg["opt"]
    ncell = args.ncell or cfg["ncell"]
    K, W = args.steps, args.warmup
    # every rank builds its own shard: a different seed = different cells, same statistics (cells never interact)
    d = domain.make_domain(ncell, opt, ntile=cfg["ntile"], glacier_top_band=cfg.get("glacier", False), seed=domain.SEED + rank)
    f, sf, dmy = domain.make_forcing(d, 0, W + K, start_doy=cfg["start_doy"])
    sd0, si0 = init_state.initial_state(d, f[0])
    m = Model(d, device=local_rank)
    m.set_state(sd0, si0)
    m.set_write_fluxes(False)          # production setting: per-cell accumulators only, no per-HRU flux table
    m.push_forcing(f, sf, dmy)
    m.synchronize()
    del f

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if W > 0:
        m.dist_prec(0, W, sync=True)
    m.reset_accum()
    barrier()
    t0 = time.perf_counter()
    m.dist_prec(W, K, sync=True)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, nlaunch = m.last_kernel_ms()
    nerr = int((m.get_cell_errors() != 0).sum())
    acc = m.get_accum()
    gather_ms = None
    if use_dist:
        # the one exchange of the path: the per-cell output table to the writer, RCCL all-gather over xGMI;
        # outside the timed region (it happens once per output step, not per model step)
        from vic_amd import shard
        barrier()
        tg = time.perf_counter()
        full = shard.gather_cell_table(acc, [ncell] * world, device=torch.device("cuda", local_rank))
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert full.shape == (acc.shape[0], ncell * world)

    if rank == 0:
        hru_per_cell = d.nhru // d.ncell
        balg = b_alg(opt, hru_per_cell)
        value = world * ncell * K / elapsed
        achieved = balg * ncell / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "cell-timesteps/s", "value": value, "unit": "cell-timesteps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"], "cells_per_gpu": ncell, "hru_per_cell": hru_per_cell,
                       "parallelism": "cells sharded across %d GPU(s), no data-path collective" % world,
                       "cells_with_error_flags": nerr, "output_gather_ms": gather_ms,
                       "mean_runoff_mm_per_step": float(acc[C["CA_RUNOFF"]].mean() / max(1, K)),
                       "mean_swe_mm_end": float(acc[C["CA_SWE_END"]].mean())},
            # one "launch" of the hot path = one model step of the rank's cells: the QUICK_FLUX path is a single kernel, the
            # finite-difference path a pipeline of kernels (stage / profile solve / surface evaluation); the duration is
            # measured with HIP events on the library's streams around the whole step
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.config, ncell),
                         "kernel": "vic_hru_step" if opt.QUICK_FLUX else "vic_fd_stage + vic_profile_solve + vic_surf_eval (per-step pipeline)",
                         "kernel_ms_per_launch": kernel_ms, "launches_timed": nlaunch,
                         "algorithmic_bytes_per_cell_step": balg,
                         "note": "fp64 VALU / divergence-bound root finding (SURVEY.md 7.3 #4): the algorithmic HBM fraction is small by construction; "
                                 "profiles/ holds the per-kernel rocprofv3 stats and PMC passes"},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # the baseline is a reported extra; never lose the bench line over it
                out["cpu_baseline"] = {"value": None, "unit": "cell-timesteps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        elif world == 1:
            out["cpu_baseline"] = {"value": None, "unit": "cell-timesteps/s", "cores": 0, "kind": "port", "sample": "skipped (--no-cpu-baseline)"}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
