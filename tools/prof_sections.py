#!/usr/bin/env python3
"""Tuning aid: builds vic_amd/libvicgpu.so with -DVIC_PROF (s_memtime section timers + trip counters inside
vic_hru_step), runs a few steps of a bench workload and prints where the wave cycles go.  The instrumented library
replaces the production one: rebuild with `python vic_amd/build.py -f` afterwards (this script does it on exit).

    python tools/prof_sections.py [--config cfg3] [--ncell 20000] [--steps 4]
"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CYC = {0: "stage / monolithic kernel total", 1: "load + prepare + aero", 2: "solve_snow", 3: "root finder (monolithic)",
       5: "compute_pot_evap", 6: "runoff", 7: "zwt + distribute_node_moisture", 8: "store", 9: "  snow_intercept (in 2)",
       10: "  snow_melt (in 2)", 11: "context get", 12: "sf_sub_post", 13: "sf_sub_pre (incl. solve_snow)", 14: "item block + context put",
       15: "sf_end (incl. runoff, zwt)",
       20: "profile kernel: gate (take items, load blocks)", 21: "profile kernel: sweeps (all node visits)", 22: "  Newton predictor loops (in 21)",
       23: "  Newton fp64 loops (in 21)", 25: "  cold-nose Brent loops (in 21)", 24: "profile kernel: finish + write record"}
CNT = {0: "waves (stage launches)", 1: "lanes", 2: "sub-steps (wave)",
       7: "SurfEB evals (wave)", 8: "SurfEB evals (lane)", 9: "SnowPackEB evals (wave)", 10: "SnowPackEB evals (lane)",
       11: "CanopyEB evals (wave)", 12: "CanopyEB evals (lane)",
       16: "lock-step: sweeps (wave)", 17: "lock-step: sweeps (lane)", 18: "lock-step: node visits (wave)",
       19: "lock-step: frozen-node visits (lane)", 22: "node visits with a Newton solve (wave)",
       20: "Newton fp64 iterations (wave)", 21: "Newton fp64 iterations (lane)",
       23: "Newton predictor iterations (wave)", 24: "Newton predictor iterations (lane)",
       25: "node-1 visits with a cold-nose Brent (wave)", 26: "cold-nose Brent solves (lane)", 27: "cold-nose Brent evaluations (lane)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--ncell", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--start-doy", type=int, default=-1)
    ap.add_argument("--prebuilt", action="store_true", help="library was already built with -DVIC_PROF (GPU box)")
    args = ap.parse_args()
    from vic_amd import build as vb
    if not args.prebuilt:
        vb.build(force=True, extra=["-DVIC_PROF"])
    import bench
    from vic_amd import domain, init_state
    from vic_amd.api import Model, load_library
    cfg = bench.config(args.config)
    opt = cfg["opt"]
    doy = cfg["start_doy"] if args.start_doy < 0 else args.start_doy
    d = domain.make_domain(args.ncell, opt, ntile=cfg["ntile"])
    f, sf, dmy = domain.make_forcing(d, 0, args.steps + 1, start_doy=doy)
    sd0, si0 = init_state.initial_state(d, f[0])
    m = Model(d, device=0)
    m.set_state(sd0, si0)
    m.set_write_fluxes(False)
    m.push_forcing(f, sf, dmy)
    lib = load_library()
    cyc = (ctypes.c_ulonglong * 32)()
    cnt = (ctypes.c_ulonglong * 32)()
    m.dist_prec(0, 1, sync=True)
    lib.vicgpu_prof_read(cyc, cnt)
    m.dist_prec(1, args.steps, sync=True)
    ms, nl = m.last_kernel_ms()
    assert lib.vicgpu_prof_read(cyc, cnt) == 0
    cyc = np.array(cyc[:], dtype=np.float64)
    cnt = np.array(cnt[:], dtype=np.float64)
    print("kernel %.3f ms/launch over %d launches (instrumented build)" % (ms, nl))
    tot = cyc[0]
    for k, name in CYC.items():
        if k >= 20:      # the profile kernel's sections: share of that kernel's own section total
            ptot = cyc[20] + cyc[21] + cyc[24]
            print("  %-50s %14.3e cyc  %5.1f %% of the profile kernel's sections" % (name, cyc[k], 100 * cyc[k] / max(ptot, 1)))
        else:
            print("  %-36s %14.0f cyc/wave  %5.1f %%" % (name, cyc[k] / max(cnt[0], 1), 100 * cyc[k] / tot))
    for k, name in CNT.items():
        print("  %-36s %14.0f" % (name, cnt[k]))
    if not args.prebuilt:
        vb.build(force=True)


if __name__ == "__main__":
    main()
