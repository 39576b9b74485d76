"""Seeded inputs for the pure-function known-answer vectors (SURVEY.md 8(c) fixture plan (i)): physically valid
ranges plus the branch points of each function.  Used by tests/golden/make_golden.py (reference outputs) and the tests."""
import numpy as np

from vic_amd.abi import C

N = 256


def pure_inputs(seed=20261003):
    rng = np.random.default_rng(seed)
    u = lambda lo, hi, n=N: rng.uniform(lo, hi, n)
    d = {}
    T = np.concatenate([u(-45, 40, N - 6), [0.0, -0.0, 1e-12, -1e-12, 40.0, -60.0]])
    d["VICGPU_PURE_SVP"] = np.stack([T], 1)
    d["VICGPU_PURE_SVP_SLOPE"] = np.stack([T], 1)
    # rain / snow partition: temperatures across both thresholds, zero and tiny precipitation
    at = np.concatenate([u(-6, 8, N - 8), [1.0, 3.0, -0.5, 0.5, 2.0, 2.999999, 1.000001, 10.0]])
    pr = np.concatenate([u(0, 30, N - 8), [0.0, 1e-6, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0]])
    d["VICGPU_PURE_CALC_RAINONLY"] = np.stack([at, pr, np.full(N, 3.0), np.full(N, 1.0)], 1)
    # snow albedo: fresh snow / accumulation / thaw branches, both last_snow regimes
    d["VICGPU_PURE_SNOW_ALBEDO"] = np.stack([
        np.where(rng.random(N) < 0.3, u(0.031, 20), np.where(rng.random(N) < 0.5, 0.0, u(0, 0.03))),   # new_snow (mm)
        np.where(rng.random(N) < 0.1, 0.0, u(0.001, 1.5)),                                              # swq (m)
        u(0.0, 3.0), u(0.3, 0.9),                                                                       # depth, albedo
        np.where(rng.random(N) < 0.5, -u(1e3, 5e6), np.where(rng.random(N) < 0.5, 0.0, u(1, 1e5))),    # cold content
        rng.choice([1.0, 3.0, 24.0], N), rng.integers(0, 200, N).astype(float), rng.integers(0, 2, N).astype(float)], 1)
    d["VICGPU_PURE_NEW_SNOW_DENSITY"] = np.stack([np.concatenate([u(-30, 5, N - 3), [0.0, -15.0, 2.5]])], 1)
    # stability correction: stable, unstable and the critical Richardson number
    d["VICGPU_PURE_STABILITY"] = np.stack([u(2, 40), u(0, 1.5), u(-30, 10), u(-30, 25), np.concatenate([u(0.1, 15, N - 2), [1e-3, 30.0]]),
                                          u(1e-4, 0.5)], 1)
    d["VICGPU_PURE_PENMAN"] = np.stack([u(-30, 35), u(0, 3500), u(-100, 700), u(0, 4000), u(2, 500), np.concatenate([u(0, 5000, N - 2), [0.0, 5000.0]]),
                                       u(0, 60)], 1)
    d["VICGPU_PURE_CALC_RC"] = np.stack([np.concatenate([u(0, 600, N - 2), [0.0, 100.0]]), u(0, 900), u(0, 120), u(-30, 40), u(0, 4500),
                                        np.concatenate([u(0.1, 8, N - 1), [0.0]]), u(1, 50), rng.integers(0, 2, N).astype(float)], 1)
    d["VICGPU_PURE_ESTIMATE_T1"] = np.stack([u(-30, 30), u(-25, 25), u(-5, 12), u(0.05, 0.2), u(0.2, 1.5), u(0.2, 3), u(0.2, 3),
                                            u(1e6, 4e6), u(2, 8), rng.choice([3600.0, 10800.0, 86400.0], N)], 1)
    moist = u(0.0, 0.45); moist[:4] = [0.0, 0.4, 0.2, 0.1]
    wu = np.where(rng.random(N) < 0.5, moist, moist * u(0, 1))
    d["VICGPU_PURE_SOIL_CONDUCTIVITY"] = np.stack([moist, wu, np.full(N, 2650.0), u(1300, 1700), np.concatenate([u(0, 0.9, N - 2), [0.19, 0.2]]),
                                                  np.full(N, 2650.0), u(1300, 1700), np.where(rng.random(N) < 0.7, 0.0, u(0, 0.3))], 1)
    d["VICGPU_PURE_VOL_HEAT_CAPACITY"] = np.stack([u(0.4, 0.7), u(0, 0.45), u(0, 0.3), u(0, 0.3)], 1)
    d["VICGPU_PURE_MAX_UNFROZEN_WATER"] = np.stack([np.concatenate([-u(1e-6, 40, N - 4), [0.0, 0.5, -1e-9, -273.0]]), u(0.3, 0.55), u(5, 40), u(4, 25)], 1)
    lx = u(-5, 5); ux = lx + u(0.01, 10)
    d["VICGPU_PURE_LINEAR_INTERP"] = np.stack([u(-6, 16), lx, ux, u(-20, 20), u(-20, 20)], 1)
    d["VICGPU_PURE_VEG_HEIGHT"] = np.stack([u(0.0, 20), np.concatenate([u(0.05, 8, N - 2), [0.0, 1e-3]])], 1)
    return {C[k]: v for k, v in d.items()}


# option sets the option-dependent functions are exercised with
OPTION_SETS = {
    "default": dict(FULL_ENERGY=1),                                                   # KIENZLE, USACE albedo, Bras density
    "alt": dict(FULL_ENERGY=1, TEMP_TH_TYPE=0, SNOW_ALBEDO=1, SNOW_DENSITY=1),        # VIC_412, SUN1999, SNTHERM
}
