#!/usr/bin/env python3
"""Benchmark of the VIC hot path (dist_prec -> full_energy -> surface_fluxes) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg4|cfg2] [--compat] [--node-solver newton|brent]

A "step" is one model time step of every cell of the rank's shard (all HRUs, all snow sub-steps) INCLUDING its put_data
(dist_prec.c:167): the aggregated output variables are updated on the device every step, and the table the writer needs
(OUT_VARS, daily aggregates as float32) is fetched once after the timed region -- that fetch, plus the all-gather of it
at N > 1, is `output_gather_ms`.
Metric: cell-timesteps/s, whole job (sum over ranks).

Node solver.  The frozen-node root finds of the soil profile run as a safeguarded Newton iteration by default (outputs
within north_star's 1e-5 of the reference: tests/test_gpu_parity.py::test_gpu_against_reference_goldens[*-newton]);
`--node-solver brent` replays the reference's Brent iteration step by step (1e-6 on every state variable).  At N = 1 the
JSON line also carries the step time of the other mode (`strict_replay_ms_per_step`), measured after the timed region.

Launching.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (one process per GPU,
`python -m torch.distributed.run --nproc-per-node N ...`, RCCL) BEFORE anything in this process touches the GPU, waits
for them and exits with their code; started under torch.distributed.run already (WORLD_SIZE set), it is one of the ranks.

Workloads (BASELINE.json configs, SURVEY.md 8(d) synthetic inputs):
  cfg3  100k cells, FULL_ENERGY + FROZEN_SOIL (10 thermal nodes, explicit), 5 snow bands x 5 veg tiles = 25 HRUs/cell,
        hourly -- the config the metric is quoted on (default at N = 1)
  cfg4  BASELINE configs[3]: the glacier domain (cfg3 + veg slot 0 of the top band a glacier HRU) sharded over the ranks,
        125k cells per GPU = 1M cells at N = 8 (default at N > 1; weak scaling: the per-GPU share is fixed).  Every rank
        builds ITS block of the one N x 125k-cell domain (domain.make_domain(cell_range=...) == shard.shard_domain of it).
  cfg2  10k cells, FULL_ENERGY (QUICK_FLUX), 1 band x 3 veg tiles, hourly
FROZEN_SOIL semantics: "fixed" (node arrays, oracle patch P2) by default, `--compat` = the reference as shipped
(frozen_soil.c:218-221 layer arrays indexed by node; SURVEY.md Finding 1.2).

Forcing for all W+K steps is generated on the host and is resident in HBM before the timed region starts.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (non-matrix) peak: 256 CUs x 128 flop/clk x 2.4 GHz
CELLS_PER_GPU_CFG4 = 125000
OUT_STEP_RATIO = 24     # hourly steps, daily output records
# the variables north_star names (runoff, baseflow, SWE, soil moisture [3 layers], glacier mass balance) + evaporation and
# precipitation: what the writer gets per output record
OUT_VARS = ["OUT_RUNOFF", "OUT_BASEFLOW", "OUT_SWE", "OUT_SOIL_MOIST", "OUT_GLAC_MBAL", "OUT_EVAP", "OUT_PREC"]


def config(name, compat=False):
    from vic_amd import abi
    fc = 1 if compat else 0
    sem = "compat" if compat else "fixed"
    if name == "cfg3":
        opt = abi.default_options(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=fc)
        return dict(opt=opt, ncell=100000, ntile=5, start_doy=60,
                    workload="cfg3: 100k cells, FULL_ENERGY+FROZEN_SOIL (Nnode=10, explicit, %s), 5 bands x 5 veg tiles, hourly" % sem)
    if name == "cfg4":       # BASELINE.json configs[3] (1M cells on 8 GPUs): cfg3 + a glacier HRU in the top band
        opt = abi.default_options(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=fc)
        return dict(opt=opt, ncell=CELLS_PER_GPU_CFG4, ntile=5, start_doy=60, glacier=True,
                    workload="cfg4: 125k cells per GPU (1M cells on 8 GPUs), FULL_ENERGY+FROZEN_SOIL (Nnode=10, explicit, %s), "
                             "5 bands x 5 veg tiles, veg slot 0 of the top band = glacier (solve_glacier / surface_fluxes_glac), hourly" % sem)
    if name == "cfg5":       # BASELINE.json configs[4]: the cfg4 domain as an I/O-overlap stress (run_cfg5)
        c = config("cfg4", compat)
        c["workload"] = ("cfg5: one GPU's 125k-cell share of the 1M-cell glacier domain (cfg4), forcing streamed as hourly raw records in "
                         "chunks, put_data every step, daily output table fetched + gathered every 24 steps inside the timed region, "
                         "state records at the end")
        return c
    if name == "cfg2":
        opt = abi.default_options(FULL_ENERGY=1)
        return dict(opt=opt, ncell=10000, ntile=3, start_doy=60,
                    workload="cfg2: 10k cells, FULL_ENERGY (QUICK_FLUX), 1 band x 3 veg tiles, hourly")
    raise SystemExit("unknown config " + name)


def b_alg(opt, hru_per_cell, n_outvar=9, out_step_ratio=OUT_STEP_RATIO):
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

    def run(ncell_s, nsteps, threads=0):
        d = domain.make_domain(ncell_s, copy.copy(opt), ntile=cfg["ntile"], glacier_top_band=cfg.get("glacier", False))
        f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=cfg["start_doy"])
        sd0, si0 = init_state.initial_state(d, f[0])
        m = pyref.RefModel(d, variant) if kind == "reference" else pyref.OracleModel(d)
        nthreads = threads or int(getattr(m.lib, m.prefix + "max_threads")())
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
    nsteps = int(min(2400, max(12, target_seconds * rate / ncell_s)))      # small domains: more steps instead of more cells
    secs, nthreads = run(ncell_s, nsteps)
    rate = ncell_s * nsteps / secs
    # the same code on ONE thread (the all-cores figure above is far from cores x this: the reference allocates per call,
    # full_energy.c:173, prepare_full_energy.c:45, and the sample is cold)
    n1 = max(16, min(256, int(3.0 * rate / max(1, nthreads) / 12)))
    secs1, _ = run(n1, 12, threads=1)
    return {"value": rate, "unit": "cell-timesteps/s", "cores": nthreads, "kind": kind,
            "sample": "%d cells x %d steps of the same workload (%.1f s, OpenMP over cells, %s)" % (
                ncell_s, nsteps, secs, "reference build oracle/_ref/libvicref_%s.so" % variant if kind == "reference" else "oracle/libvicoracle.so"),
            "one_thread_value": n1 * 12 / secs1, "one_thread_sample": "%d cells x 12 steps, 1 thread (%.1f s)" % (n1, secs1)}


def csrc_digest():
    """Content hash of the device sources + ABI headers: the build a committed profile belongs to (the GPU box has no .git)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vic_amd", "csrc")
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(d)) + [os.path.join(inc, x) for x in sorted(os.listdir(inc))]:
        with open(fn if os.path.isabs(fn) else os.path.join(d, fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(config_name, ncell, compat):
    """HBM bytes per step from the committed PMC passes of this workload (profiles/traffic_<config>.json, written by
    tools/round_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command).  The file is stamped with the
    digest of the device sources it was measured on; a stale stamp (any kernel source changed since) gives None."""
    path = os.path.join(ROOT, "profiles", "traffic_%s%s.json" % (config_name, "_compat" if compat else ""))
    if not os.path.exists(path):
        return None, "no PMC measurement committed for this workload"
    with open(path) as f:
        t = json.load(f)
    if t.get("config") != config_name or t.get("cells_per_gpu") != ncell:
        return None, "committed PMC measurement is for another size"
    if t.get("csrc_digest") != csrc_digest():
        return None, "committed PMC measurement is stale (device sources changed since %s)" % t.get("round", "?")
    return t.get("hbm_bytes_per_step"), "profiles/%s: %s" % (os.path.basename(path), t.get("method", ""))


def valu_roofline(config_name, cell_steps_per_s):
    """Secondary roofline (the path is bound by divergent fp64 arithmetic, not by HBM): the fp64 operations the
    reference's algorithm executes per cell-step, counted on the CPU restatement (profiles/flops_<config>.json, written by
    tools/count_flops.py: adds, multiplies, divisions and square roots counted one each, libm calls listed beside them) times
    the measured cell-steps/s, against the fp64 vector peak; plus the lanes active per vector instruction from the committed
    PMC pass (profiles/pmc_<config>.json), when there is one."""
    path = os.path.join(ROOT, "profiles", "flops_%s.json" % config_name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        fl = json.load(f)
    ach = fl["fp64_ops_per_cell_step"] * cell_steps_per_s / 1e12
    out = {"bound": "fp64 VALU", "counted_fp64_ops_per_cell_step": fl["fp64_ops_per_cell_step"], "libm_calls_per_cell_step": fl["libm_calls_per_cell_step"],
           "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
           "libm_calls_per_s": fl["libm_calls_per_cell_step"] * cell_steps_per_s, "source": "profiles/flops_%s.json (%s)" % (config_name, fl.get("sample", "")),
           "note": "operations of the REFERENCE's algorithm (its Brent iterations and Gauss-Seidel sweeps); a division, square root or libm call "
                   "counts as one operation here although it costs ~10-100 vector instructions"}
    pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % config_name)
    if os.path.exists(pmc):
        with open(pmc) as f:
            pj = json.load(f)
        out["lanes_active_per_valu_inst"] = pj.get("lanes_active_per_valu_inst")
        out["valu_active_share_of_wave_cycles"] = pj.get("valu_active_share_of_wave_cycles")
        out["pmc_source"] = "profiles/pmc_%s.json (%s)" % (config_name, pj.get("round", ""))
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(ngpus, argv):
    """One process per GPU through torch.distributed.run; this (parent) process never initialises the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def launch_check(args):
    """`--launch-check`: the ranks only rendezvous (gloo, no GPU) and report who is there -- the CPU test of the launcher."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    pids = [None] * world
    dist.all_gather_object(pids, os.getpid())
    if rank == 0:
        print(json.dumps({"launch_check": True, "gpus_requested": args.gpus, "ranks_seen": world, "pids": pids}))
    dist.barrier()
    dist.destroy_process_group()


RAW_FROM_TABLE = (("VIC_RAW_AIR_TEMP", "VIC_F_AIR_TEMP", 1.0), ("VIC_RAW_PREC", "VIC_F_PREC", 1.0), ("VIC_RAW_PRESSURE_KPA", "VIC_F_PRESSURE", 1e-3),
                  ("VIC_RAW_VP_KPA", "VIC_F_VP", 1e-3), ("VIC_RAW_SHORTWAVE", "VIC_F_SHORTWAVE", 1.0), ("VIC_RAW_LONGWAVE", "VIC_F_LONGWAVE", 1.0),
                  ("VIC_RAW_WIND", "VIC_F_WIND", 1.0))


def cfg5_sequence(m, f, dmy, step0, nsteps, chunk, out_every, opt, gather, bufs=None):
    """The cfg5 step sequence on model `m` (vicNl.c:506-610 with the forcing arriving in chunks): steps [step0, step0 + nsteps) of
    the derived table `f`, handed over as hourly RAW records chunk by chunk (vicgpu_prefetch_forcing_raw: upload + derivation of
    atmos[rec] on the device while the previous chunk runs), put_data inside every step, and every `out_every` steps the writer's
    table fetched and passed to `gather`.  Returns the list of gathered tables.  Used by bench.py (timed) and by
    tests/test_gpu_parity.py::test_cfg5_sequence (checked)."""
    import numpy as np
    from vic_amd.abi import C
    assert nsteps % chunk == 0 and out_every % chunk == 0
    ncell = f.shape[-1]
    if bufs is None:
        bufs = [m.pinned((chunk, C["VIC_NRAW"], opt.dt, ncell)) for _ in range(2)]

    def fill(buf, lo):
        fs = f[lo:lo + chunk]
        for name, src, scale in RAW_FROM_TABLE:
            buf[:, C[name]] = fs[:, C[src], :opt.NF] * scale
    nch = nsteps // chunk
    tables = []
    fill(bufs[0], step0)
    m.prefetch_forcing_raw(bufs[0], dmy[step0:step0 + chunk]); m.swap_forcing()
    for k in range(nch):
        if k + 1 < nch:
            lo = step0 + (k + 1) * chunk
            fill(bufs[(k + 1) % 2], lo)                                      # the host prepares the next chunk ...
            m.prefetch_forcing_raw(bufs[(k + 1) % 2], dmy[lo:lo + chunk])    # ... and starts its upload + derivation
        m.dist_prec(0, chunk, sync=False)
        if ((k + 1) * chunk) % out_every == 0:
            tables.append(gather(m.get_outputs(OUT_VARS, reset=True)))       # one output record leaves the device (waits for its steps)
        if k + 1 < nch:
            m.swap_forcing()
    m.synchronize()
    return tables


def run_cfg5(args, cfg, d, f, sf, dmy, sd0, si0, world, rank, local_rank, use_dist, ncell, ncell_global):
    """`--config cfg5`: K steps of cfg5_sequence in the timed region, the same K steps with the forcing resident beside it, the
    upload + derivation of one chunk on its own (how much of it the overlap hides), and the state records at the end."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vic_amd import shard
    from vic_amd.api import Model
    opt = cfg["opt"]
    K, W, CH = args.steps, args.warmup, 6
    K = max(OUT_STEP_RATIO, (K // OUT_STEP_RATIO) * OUT_STEP_RATIO)             # whole output records
    W = (W // CH) * CH

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    gather_ms = []

    def gather(outs):
        t0 = time.perf_counter()
        full = shard.gather_cell_table(outs, [ncell] * world, device=torch.device("cuda", local_rank), root=0) if use_dist else outs
        gather_ms.append((time.perf_counter() - t0) * 1e3)
        return full

    def make():
        m = Model(d, device=local_rank)
        m.set_state(sd0, si0); m.set_write_fluxes(False); m.put_data_config(OUT_STEP_RATIO); m.put_data_init()
        return m
    # resident leg: the same steps with the whole forcing table already in HBM
    m0 = make()
    raw_all = np.zeros((W + K, C_()["VIC_NRAW"], opt.dt, ncell))
    for name, src, scale in RAW_FROM_TABLE:
        raw_all[:, C_()[name]] = f[:W + K, C_()[src], :opt.NF] * scale
    m0.prefetch_forcing_raw(raw_all, dmy[:W + K]); m0.swap_forcing(); m0.synchronize()       # the same records, derived at once
    del raw_all
    if W:
        m0.dist_prec(0, W, sync=True)
        m0.get_outputs(OUT_VARS, reset=True)
    barrier(); t0 = time.perf_counter()
    for k in range(K // OUT_STEP_RATIO):
        m0.dist_prec(W + k * OUT_STEP_RATIO, OUT_STEP_RATIO, sync=False)
        m0.get_outputs(OUT_VARS, reset=True)
    m0.synchronize(); barrier()
    resident_ms = (time.perf_counter() - t0) / K * 1e3
    state_ref = m0.get_state()
    m0.close()
    # streamed leg (the timed one)
    m = make()
    bufs = [m.pinned((CH, C_()["VIC_NRAW"], opt.dt, ncell)) for _ in range(2)]
    if W:
        cfg5_sequence(m, f, dmy, 0, W, CH, W, opt, lambda o: o, bufs)
        m.get_outputs(OUT_VARS, reset=True)
    m.reset_accum()
    barrier(); t0 = time.perf_counter()
    tables = cfg5_sequence(m, f, dmy, W, K, CH, OUT_STEP_RATIO, opt, gather, bufs)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, nlaunch = m.last_kernel_ms()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(m.get_state(), state_ref))
    nerr = int((m.get_cell_errors() != 0).sum())
    # one chunk's upload + derivation on its own
    fs = f[W:W + CH]
    for name, src, scale in RAW_FROM_TABLE:
        bufs[0][:, C_()[name]] = fs[:, C_()[src], :opt.NF] * scale
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.prefetch_forcing_raw(bufs[0], dmy[W:W + CH]); m.swap_forcing(); m.synchronize()
    upload_ms = (time.perf_counter() - t0) / CH * 1e3
    # state save: the records in state-file order (write_model_state.c:95-337) to the host
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec = m.get_state_records()
    state_save_ms = (time.perf_counter() - t0) * 1e3
    m.close()
    streamed_ms = elapsed / K * 1e3
    hidden = 1.0 - max(0.0, streamed_ms - resident_ms) / upload_ms if upload_ms > 0 else None
    if rank != 0:
        return None
    hru_per_cell = d.nhru // d.ncell
    balg = b_alg(opt, hru_per_cell)
    achieved = balg * ncell / (streamed_ms * 1e-3) / 1e9
    return {
        "metric": "cell-timesteps/s", "value": world * ncell * K / elapsed, "unit": "cell-timesteps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": streamed_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["workload"], "cells_per_gpu": ncell, "cells_total": ncell_global, "hru_per_cell": hru_per_cell,
                   "node_solver": args.node_solver, "forcing_chunk_steps": CH, "out_step_ratio": OUT_STEP_RATIO,
                   "resident_forcing_ms_per_step": resident_ms, "chunk_upload_and_derive_ms_per_step": upload_ms, "h2d_hidden_frac": hidden,
                   "output_records_gathered": len(tables), "output_gather_ms": float(np.mean(gather_ms)) if gather_ms else None,
                   "output_table": "%s as float32 [%d][%d]" % (",".join(OUT_VARS), tables[0].shape[0], tables[0].shape[1]) if tables and tables[0] is not None else None,
                   "state_save_ms": state_save_ms, "state_record_bytes": int(rec.nbytes), "streamed_state_equals_resident": bool(same),
                   "cells_with_error_flags": nerr,
                   "parallelism": "cells sharded across %d GPU(s), output table gathered to rank 0 every %d steps" % (world, OUT_STEP_RATIO)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "whole cfg5 step (pipeline + put_data + streamed forcing)", "kernel_ms_per_launch": streamed_ms,
                     "algorithmic_bytes_per_cell_step": balg, "csrc_digest": csrc_digest()},
        "cpu_baseline": {"value": None, "unit": "cell-timesteps/s", "cores": 0, "kind": "port", "sample": "not timed for cfg5 (see the cfg3 line)"},
    }


def C_():
    from vic_amd.abi import C
    return C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default=None, help="cfg3 (default at 1 GPU), cfg4 (default at > 1 GPU), cfg2, cfg5 (I/O-overlap stress on the cfg4 domain)")
    ap.add_argument("--compat", action="store_true", help="FROZEN_SOIL as the reference ships it (frozen_soil.c:218-221) instead of 'fixed'")
    ap.add_argument("--node-solver", default="newton", choices=["newton", "brent"],
                    help="frozen-node root finder (vicgpu_options.NODE_SOLVER): converged Newton (default) or the reference's Brent iteration replayed")
    ap.add_argument("--no-strict-leg", action="store_true", help="skip the second timing with the other node solver (N = 1)")
    ap.add_argument("--no-stream-leg", action="store_true", help="skip the PCIe-inclusive timing (forcing streamed in chunks, N = 1)")
    ap.add_argument("--no-compat-leg", action="store_true", help="skip the 4-step timing of frozen_compat = 1 (the reference's FROZEN_SOIL as shipped, N = 1)")
    ap.add_argument("--ncell", type=int, default=0, help="override cells per GPU (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))        # nothing above has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: launch with `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.launch_check:
        return launch_check(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ranks_seen = dist.get_world_size() if use_dist else 1
    assert ranks_seen == world

    from vic_amd import domain, init_state, shard
    from vic_amd.api import Model
    from vic_amd.abi import C

    cfg_name = args.config or ("cfg3" if world == 1 else "cfg4")
    cfg = config(cfg_name, compat=args.compat)
    opt = cfg["opt"]
    opt.NODE_SOLVER = C["VIC_NODE_SOLVER_NEWTON"] if args.node_solver == "newton" else C["VIC_NODE_SOLVER_BRENT"]
    ncell = args.ncell or cfg["ncell"]
    K, W = args.steps, args.warmup
    if cfg_name == "cfg5":                # whole output records and whole forcing chunks
        K = args.steps = max(OUT_STEP_RATIO, (K // OUT_STEP_RATIO) * OUT_STEP_RATIO)
        W = args.warmup = (W // 6) * 6
    # ONE domain of world x ncell cells, cut into contiguous HRU-balanced blocks (shard.partition_cells: every cell has the
    # same number of HRUs here, so the blocks are equal); this rank builds only its block
    ncell_global = ncell * world
    c0, c1 = rank * ncell, (rank + 1) * ncell
    d = domain.make_domain(ncell_global, opt, ntile=cfg["ntile"], glacier_top_band=cfg.get("glacier", False), cell_range=(c0, c1))
    f, sf, dmy = domain.make_forcing(d, 0, W + K, start_doy=cfg["start_doy"])
    sd0, si0 = init_state.initial_state(d, f[0])
    if cfg.get("glacier"):
        # the driver opens the glacier mass-balance accumulation window (accumulateGlacierMassBalance.c:13-67)
        isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
        sd0[C["SD_GLAC_CUM_MASS_BALANCE"], isg] = 0.0
    if cfg_name == "cfg5":
        out = run_cfg5(args, cfg, d, f, sf, dmy, sd0, si0, world, rank, local_rank, use_dist, ncell, ncell_global)
        if rank == 0:
            print(json.dumps(out))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    def make_model(dom):
        mm = Model(dom, device=local_rank)
        mm.set_state(sd0, si0)
        mm.set_write_fluxes(False)     # production setting: nobody reads the per-HRU flux table on the host ...
        mm.put_data_config(OUT_STEP_RATIO)   # ... put_data does, on the device (this turns its rows back on inside the step)
        mm.put_data_init()
        mm.push_forcing(f, sf, dmy)
        mm.synchronize()
        return mm

    m = make_model(d)

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
    # the writer's table: the aggregates of OUT_VARS as float32 [rows][cells] (WriteOutputNetCDF.c:387-455), fetched once per
    # output record -- outside the timed region (it happens every OUT_STEP_RATIO steps, not per model step)
    barrier()
    tg = time.perf_counter()
    outs = m.get_outputs(OUT_VARS, reset=True)
    full = outs
    if use_dist:
        # the one exchange of the path: that table to the writer, RCCL all-gather over xGMI
        full = shard.gather_cell_table(outs, [ncell] * world, device=torch.device("cuda", local_rank))
    barrier()
    gather_ms = (time.perf_counter() - tg) * 1e3
    assert full.shape == (outs.shape[0], ncell_global) and full.dtype == np.float32
    assert np.array_equal(full[:, c0:c1], outs, equal_nan=True)              # this rank's block arrived where the writer expects it
    if use_dist:
        nerr_t = torch.tensor([nerr], device="cuda", dtype=torch.int64)
        dist.all_reduce(nerr_t)
        nerr = int(nerr_t.item())
    # the same K steps with the other node solver (N = 1): what the choice of solver costs / buys
    other_ms = None
    if world == 1 and not opt.QUICK_FLUX and not args.no_strict_leg:
        import copy
        d2 = copy.copy(d)
        d2.opt = copy.copy(opt)
        d2.opt.NODE_SOLVER = C["VIC_NODE_SOLVER_BRENT"] if args.node_solver == "newton" else C["VIC_NODE_SOLVER_NEWTON"]
        m.close()
        m2 = make_model(d2)
        if W > 0:
            m2.dist_prec(0, W, sync=True)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        m2.dist_prec(W, K, sync=True)
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - ts) / K * 1e3
        m2.close()
    # FROZEN_SOIL exactly as the reference ships it (frozen_soil.c:218-221, SURVEY.md Finding 1.2 / 8(d) "both reported"): a few
    # steps of the same workload with frozen_compat = 1
    compat_ms = None
    if world == 1 and opt.FROZEN_SOIL and not opt.frozen_compat and not args.no_compat_leg:
        import copy
        d4 = copy.copy(d)
        d4.opt = copy.copy(opt)
        d4.opt.frozen_compat = 1
        m4 = make_model(d4)
        m4.dist_prec(0, 1, sync=True)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        m4.dist_prec(1, 4, sync=True)
        torch.cuda.synchronize()
        compat_ms = (time.perf_counter() - ts) / 4 * 1e3
        m4.close()
    # PCIe-inclusive rate (N = 1): the same K steps with the forcing arriving as hourly RAW values in chunks of 6 steps from
    # pinned host memory, each chunk uploading (and derived on the device) while the previous one runs
    stream_ms = None
    if world == 1 and not args.no_stream_leg and opt.snow_step == 1:
        m3 = Model(d, device=local_rank)
        m3.set_state(sd0, si0); m3.set_write_fluxes(False); m3.put_data_config(OUT_STEP_RATIO); m3.put_data_init()
        CH = 6
        nch = max(1, K // CH)
        bufs = [m3.pinned((CH, C["VIC_NRAW"], opt.dt, ncell)) for _ in range(2)]

        def fill(buf, lo):
            fs = f[lo:lo + CH]
            for name, src, scale in (("VIC_RAW_AIR_TEMP", "VIC_F_AIR_TEMP", 1.0), ("VIC_RAW_PREC", "VIC_F_PREC", 1.0),
                                     ("VIC_RAW_PRESSURE_KPA", "VIC_F_PRESSURE", 1e-3), ("VIC_RAW_VP_KPA", "VIC_F_VP", 1e-3),
                                     ("VIC_RAW_SHORTWAVE", "VIC_F_SHORTWAVE", 1.0), ("VIC_RAW_LONGWAVE", "VIC_F_LONGWAVE", 1.0),
                                     ("VIC_RAW_WIND", "VIC_F_WIND", 1.0)):
                buf[:, C[name]] = fs[:, C[src], :opt.NF] * scale
        fill(bufs[0], W)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        m3.prefetch_forcing_raw(bufs[0], dmy[W:W + CH]); m3.swap_forcing()
        for k in range(nch):
            if k + 1 < nch:
                fill(bufs[(k + 1) % 2], W + (k + 1) * CH)           # the host prepares the next chunk ...
                m3.prefetch_forcing_raw(bufs[(k + 1) % 2], dmy[W + (k + 1) * CH:W + (k + 2) * CH])   # ... and starts its upload
            m3.dist_prec(0, CH, sync=False)
            if k + 1 < nch:
                m3.swap_forcing()
        m3.synchronize()
        stream_ms = (time.perf_counter() - ts) / (nch * CH) * 1e3
        m3.close()
    del f

    if rank == 0:
        hru_per_cell = d.nhru // d.ncell
        balg = b_alg(opt, hru_per_cell)
        value = world * ncell * K / elapsed
        achieved = balg * ncell / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_note = measured_traffic(cfg_name, ncell, args.compat)
        row = {n: i for i, n in enumerate(["OUT_RUNOFF", "OUT_BASEFLOW", "OUT_SWE"])}
        out = {
            "metric": "cell-timesteps/s", "value": value, "unit": "cell-timesteps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"], "cells_per_gpu": ncell, "cells_total": ncell_global, "hru_per_cell": hru_per_cell,
                       "frozen_soil_semantics": ("compat" if opt.frozen_compat else "fixed") if opt.FROZEN_SOIL else None,
                       "parallelism": "cells sharded across %d GPU(s) (one process per GPU, contiguous blocks of one domain), no data-path collective" % world,
                       "ranks_seen_by_rccl": ranks_seen if use_dist else None,
                       "node_solver": args.node_solver,
                       ("strict_replay_ms_per_step" if args.node_solver == "newton" else "newton_ms_per_step"): other_ms,
                       "streamed_raw_forcing_ms_per_step": stream_ms,
                       "compat_ms_per_step": compat_ms,
                       "put_data": "on device every step (vic_put_sum / _finish / _aggregate), out_step_ratio %d" % OUT_STEP_RATIO,
                       "output_table": "%s as float32 [%d][%d]" % (",".join(OUT_VARS), full.shape[0], full.shape[1]),
                       "cells_with_error_flags": nerr, "output_gather_ms": gather_ms,
                       "mean_runoff_mm_per_step": float(full[row["OUT_RUNOFF"]].mean() / max(1, K)),
                       "mean_swe_mm_end": float(full[row["OUT_SWE"]].mean())},
            # one "launch" of the hot path = one model step of the rank's cells: the QUICK_FLUX path is a single kernel, the
            # finite-difference path a pipeline of kernels (stage / profile solve / surface evaluation); the duration is
            # measured with HIP events on the library's streams around the whole step
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "vic_hru_step" if opt.QUICK_FLUX else "vic_fd_stage + vic_profile_solve + vic_surf_eval (per-step pipeline)",
                         "kernel_ms_per_launch": kernel_ms, "launches_timed": nlaunch,
                         "algorithmic_bytes_per_cell_step": balg, "csrc_digest": csrc_digest(),
                         "note": "fp64 VALU / divergence-bound root finding (SURVEY.md 7.3 #4): the algorithmic HBM fraction is small by construction; "
                                 "profiles/ holds the per-kernel rocprofv3 stats and PMC passes"},
        }
        out["roofline_valu"] = valu_roofline(cfg_name, value / world)       # per GPU
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
