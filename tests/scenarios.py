"""Option-branch scenarios shared by the oracle-vs-reference tests (tests/test_oracle.py) and the GPU parity tests
(tests/test_gpu_parity.py): one entry per run-time option branch of the path (vic_amd/csrc/vic_types.hpp Opt,
SURVEY.md Appendix B), so that no branch is implemented without being pinned oracle-vs-reference AND device-vs-oracle.

Each entry: name -> dict(kw=option overrides, variant=reference build (plain/fixed/compat), ncell, ntile, glacier,
nsteps (side-by-side run length), doy (start day of year), tweak=optional name of a domain/forcing modifier below).
"""
import numpy as np

from vic_amd import abi, domain
from vic_amd.abi import C

FROZEN = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, frozen_compat=0)

OPTION_BRANCHES = {
    # finite-difference soil heat variants (frozen_soil.c:181-212)
    "frozen_exp_trans": dict(kw=dict(FROZEN, EXP_TRANS=1), variant="fixed", ncell=4, ntile=2, nsteps=150, doy=330),
    "frozen_exp_trans_noflux": dict(kw=dict(FROZEN, EXP_TRANS=1, NOFLUX=1), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=10),
    "frozen_noflux": dict(kw=dict(FROZEN, NOFLUX=1), variant="fixed", ncell=4, ntile=2, nsteps=120, doy=1),
    "frozen_n12": dict(kw=dict(FROZEN, Nnode=12), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=300),
    "frozen_n18": dict(kw=dict(FROZEN, Nnode=18), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=20),
    "frozen_n5": dict(kw=dict(FROZEN, Nnode=5), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=320),
    "frozen_n24": dict(kw=dict(FROZEN, Nnode=24), variant="fixed", ncell=4, ntile=2, nsteps=80, doy=25),
    # ground heat flux forms (func_surf_energy_bal.c:176-182, 234-276)
    "gf_406": dict(kw=dict(FULL_ENERGY=1, GRND_FLUX_TYPE=C["VIC_GF_406"]), variant="plain", ncell=6, ntile=3, nsteps=200, doy=80),
    "gf_full": dict(kw=dict(FULL_ENERGY=1, GRND_FLUX_TYPE=C["VIC_GF_FULL"]), variant="plain", ncell=6, ntile=3, nsteps=200, doy=80),
    "gf_406_frozen": dict(kw=dict(FROZEN, GRND_FLUX_TYPE=C["VIC_GF_406"]), variant="fixed", ncell=4, ntile=3, nsteps=120, doy=330),
    "gf_full_frozen": dict(kw=dict(FROZEN, GRND_FLUX_TYPE=C["VIC_GF_FULL"]), variant="fixed", ncell=4, ntile=3, nsteps=120, doy=330),
    # canopy-snow aerodynamic resistance variants (func_canopy_energy_bal.c:50-95): winter, overstory tiles
    "ar_406": dict(kw=dict(FULL_ENERGY=1, AERO_RESIST_CANSNOW=C["VIC_AR_406"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=350),
    "ar_406_ls": dict(kw=dict(FULL_ENERGY=1, AERO_RESIST_CANSNOW=C["VIC_AR_406_LS"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=350),
    "ar_410": dict(kw=dict(FULL_ENERGY=1, AERO_RESIST_CANSNOW=C["VIC_AR_410"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=350),
    "ar_combo": dict(kw=dict(FULL_ENERGY=1, AERO_RESIST_CANSNOW=C["VIC_AR_COMBO"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=350),
    # snow parameterisations (snow_utility.c:9-307, calc_rainonly.c:56-95)
    "snthrm": dict(kw=dict(FULL_ENERGY=1, SNOW_DENSITY=C["VIC_DENS_SNTHRM"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=1),
    "sun1999": dict(kw=dict(FULL_ENERGY=1, SNOW_ALBEDO=C["VIC_SNOW_ALBEDO_SUN1999"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=60),
    "vic412": dict(kw=dict(FULL_ENERGY=1, TEMP_TH_TYPE=C["VIC_TEMP_TH_VIC_412"]), variant="plain", ncell=6, ntile=3, nsteps=240, doy=80),
    "snthrm_frozen": dict(kw=dict(FROZEN, SNOW_DENSITY=C["VIC_DENS_SNTHRM"], SNOW_ALBEDO=C["VIC_SNOW_ALBEDO_SUN1999"]), variant="fixed",
                          ncell=4, ntile=3, nsteps=100, doy=20),
    # solver error handling: TFALLBACK off, undisturbed and with forcing that makes the root finders fail
    "tfallback0": dict(kw=dict(FULL_ENERGY=1, TFALLBACK=0), variant="plain", ncell=6, ntile=3, nsteps=200, doy=70),
    "tfallback0_frozen": dict(kw=dict(FROZEN, TFALLBACK=0), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=330),
    "stress_fallback": dict(kw=dict(FULL_ENERGY=1, TFALLBACK=1), variant="plain", ncell=8, ntile=3, nsteps=96, doy=70, tweak="stress"),
    "stress_fallback_frozen": dict(kw=dict(FROZEN, TFALLBACK=1), variant="fixed", ncell=6, ntile=2, nsteps=72, doy=330, tweak="stress"),
    "stress_error": dict(kw=dict(FULL_ENERGY=1, TFALLBACK=0), variant="plain", ncell=8, ntile=3, nsteps=96, doy=70, tweak="stress",
                         expect_errors=True),
    # GLACIER_DYNAMICS: zero-area glacier HRUs still run (full_energy.c:220, 389)
    "glacier_dynamics": dict(kw=dict(FULL_ENERGY=1, Nband=3, GLACIER_DYNAMICS=1), variant="plain", ncell=6, ntile=2, glacier=True,
                             nsteps=200, doy=120, tweak="zero_area_glacier"),
    "glacier_dynamics_frozen": dict(kw=dict(FROZEN, Nband=2, GLACIER_DYNAMICS=1), variant="fixed", ncell=4, ntile=2, glacier=True,
                                    nsteps=100, doy=100, tweak="zero_area_glacier"),
    # BLOWING: sublimation from blowing snow (CalcBlowingSnow.c), once per snow sub-step for HRUs without overstory and for
    # glacier HRUs; strong wind, so that the saltation threshold is exceeded in part of the wind distribution
    "blowing": dict(kw=dict(FULL_ENERGY=1, BLOWING=1), variant="plain", ncell=6, ntile=3, nsteps=160, doy=350, tweak="windy"),
    "blowing_frozen": dict(kw=dict(FROZEN, BLOWING=1), variant="fixed", ncell=4, ntile=2, nsteps=80, doy=20, tweak="windy"),
    "blowing_glacier": dict(kw=dict(FULL_ENERGY=1, Nband=3, BLOWING=1), variant="plain", ncell=4, ntile=2, glacier=True, nsteps=120, doy=30,
                            tweak="windy"),
    "blowing_wb_daily": dict(kw=dict(FULL_ENERGY=0, dt=24, snow_step=3, BLOWING=1), variant="plain", ncell=6, ntile=2, nsteps=40, doy=340,
                             tweak="windy"),
}

# IMPLICIT soil heat solution (newt_raph_func_fast.c, frozen_soil.c:229-301,540-803): Newton iteration with the explicit
# solver as its fallback.  "fixed" reference variant = P1 + P2 + P3 (oracle/ref_build/build_ref.sh).
IMPLICIT_BRANCHES = {
    # Order matters in one process: the patched reference keeps fda_heat_eqn's work arrays in thread-local statics, and the
    # bottom node reads kappa_new[n + 1], which nothing ever assigns -- 0 in any real run (n is fixed), but a left-over of an
    # earlier, larger n in a test process.  So the node count never decreases along this list.
    "implicit": dict(kw=dict(FROZEN, IMPLICIT=1), variant="fixed", ncell=4, ntile=2, nsteps=150, doy=330),
    "implicit_spring": dict(kw=dict(FROZEN, IMPLICIT=1), variant="fixed", ncell=4, ntile=3, nsteps=120, doy=95),
    "implicit_exp_trans": dict(kw=dict(FROZEN, IMPLICIT=1, EXP_TRANS=1), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=330),
    "implicit_glacier": dict(kw=dict(FROZEN, IMPLICIT=1, Nband=2), variant="fixed", ncell=4, ntile=2, glacier=True, nsteps=80, doy=100),
    "implicit_noflux_n12": dict(kw=dict(FROZEN, IMPLICIT=1, NOFLUX=1, Nnode=12), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=10),
    # the implicit solver's own limit (newt_raph_func_fast.c:7: 21 nodes)
    "implicit_n21": dict(kw=dict(FROZEN, IMPLICIT=1, Nnode=21), variant="fixed", ncell=4, ntile=2, nsteps=80, doy=330),
}

# QUICK_SOLVE (calc_surf_energy_bal.c:289-309, 400-480): the Tsurf iteration on the nodes above the thaw depth + 4, a second
# iteration on the whole column when the surface changes sign, the final evaluation on the whole column
QUICK_SOLVE_BRANCHES = {
    "quick_solve": dict(kw=dict(FROZEN, QUICK_SOLVE=1), variant="fixed", ncell=4, ntile=3, nsteps=150, doy=95),
    "quick_solve_winter": dict(kw=dict(FROZEN, QUICK_SOLVE=1, Nnode=12), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=330),
    "quick_solve_glacier": dict(kw=dict(FROZEN, QUICK_SOLVE=1, Nband=2), variant="fixed", ncell=4, ntile=2, glacier=True, nsteps=80, doy=100),
    # with NOFLUX / EXP_TRANS: the iteration runs with both forced off, NOFLUX returns with the second iteration only, EXP_TRANS
    # not at all (calc_surf_energy_bal.c:298-309, 403)
    "quick_solve_noflux": dict(kw=dict(FROZEN, QUICK_SOLVE=1, NOFLUX=1), variant="fixed", ncell=4, ntile=3, nsteps=150, doy=95),
    "quick_solve_exp_trans": dict(kw=dict(FROZEN, QUICK_SOLVE=1, EXP_TRANS=1), variant="fixed", ncell=4, ntile=2, nsteps=100, doy=95),
    "quick_solve_noflux_exp_trans_n12": dict(kw=dict(FROZEN, QUICK_SOLVE=1, NOFLUX=1, EXP_TRANS=1, Nnode=12), variant="fixed", ncell=4, ntile=2,
                                             nsteps=100, doy=330),
}


def build(name, nsteps=None):
    """Domain + forcing of a scenario: returns (spec, d, f, sf, dmy)."""
    sp = OPTION_BRANCHES.get(name) or IMPLICIT_BRANCHES.get(name) or QUICK_SOLVE_BRANCHES.get(name) or RANDOM_COMBINATIONS[name]
    opt = abi.default_options(**sp["kw"])
    d = domain.make_domain(sp["ncell"], opt, ntile=sp["ntile"], glacier_top_band=sp.get("glacier", False))
    n = nsteps or sp["nsteps"]
    f, sf, dmy = domain.make_forcing(d, 0, n, start_doy=sp["doy"])
    tw = sp.get("tweak")
    if tw == "stress":
        # every 7th step one third of the cells receive a shortwave flux no surface temperature within the solvers'
        # search range (+-51 K, root_brent.c:183-248) can balance: the root finders fail -> fallback (or ERROR)
        sw = C["VIC_F_SHORTWAVE"]
        cells = np.arange(d.ncell) % 3 == 1
        for s in range(5, n, 7):
            f[s, sw][:, cells] = 60000.0
    elif tw == "windy":
        f[:, C["VIC_F_WIND"]] *= 3.5
    elif tw == "zero_area_glacier":
        # glacier HRUs of every second cell lose their area (Cv = 0): skipped without GLACIER_DYNAMICS, run with it
        isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
        cell = d.hru_iparams[C["HPI_CELL"]]
        d.hru_dparams[C["HPD_CV"], isg & (cell % 2 == 0)] = 0.0
    return sp, d, f, sf, dmy


def random_combinations(n=14, seed=2026):
    """Seeded random combinations of the run-time options (valid ones: get_global_param.c:376-381, 1151-1155 and what
    vicgpu_create accepts), so that interactions between branches are pinned too, not only each branch on its own.
    Returns [(name, spec)] in the format of OPTION_BRANCHES."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        frozen = bool(rng.integers(0, 2)) or k % 3 == 0
        kw = dict(FULL_ENERGY=1)
        if frozen:
            kw.update(FROZEN_SOIL=1, frozen_compat=0, Nnode=int(rng.choice([5, 8, 10, 12, 18])), NOFLUX=int(rng.integers(0, 2)),
                      EXP_TRANS=int(rng.integers(0, 2)))
            mode = rng.integers(0, 4)
            if mode == 1 and kw["Nnode"] <= 18:
                kw["IMPLICIT"] = 1
            elif mode == 2 and not kw["NOFLUX"] and not kw["EXP_TRANS"]:
                kw["QUICK_SOLVE"] = 1
        elif rng.integers(0, 4) == 0:
            kw = dict(FULL_ENERGY=0, dt=int(rng.choice([3, 24])), snow_step=3)
        kw["Nband"] = int(rng.integers(1, 4))
        kw["GRND_FLUX_TYPE"] = int(rng.choice([C["VIC_GF_406"], C["VIC_GF_410"], C["VIC_GF_FULL"]]))
        kw["AERO_RESIST_CANSNOW"] = int(rng.choice([C["VIC_AR_406"], C["VIC_AR_406_LS"], C["VIC_AR_406_FULL"], C["VIC_AR_410"], C["VIC_AR_COMBO"]]))
        kw["SNOW_DENSITY"] = int(rng.integers(0, 2)); kw["SNOW_ALBEDO"] = int(rng.integers(0, 2)); kw["TEMP_TH_TYPE"] = int(rng.integers(0, 2))
        kw["CORRPREC"] = int(rng.integers(0, 2)); kw["TFALLBACK"] = int(rng.integers(0, 4) > 0)
        glacier = bool(rng.integers(0, 3) == 0) and kw["Nband"] > 1
        if glacier:
            kw["GLACIER_DYNAMICS"] = int(rng.integers(0, 2))
        out.append(("combo%02d" % k, dict(kw=kw, variant="fixed" if kw.get("FROZEN_SOIL") else "plain", ncell=5, ntile=2, glacier=glacier,
                                          nsteps=24, doy=int(rng.choice([20, 95, 200, 330])))))
    return out


def random_combinations_round3(n=10, seed=0x516D):
    """A second batch for the options that arrived in round 3: BLOWING on any surface, QUICK_SOLVE with any NOFLUX / EXP_TRANS,
    node counts up to 24 (IMPLICIT up to its own limit of 21), all drawn together with the older switches."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        frozen = bool(rng.integers(0, 3)) or k % 2 == 0
        kw = dict(FULL_ENERGY=1, BLOWING=int(rng.integers(0, 3) > 0))
        if frozen:
            kw.update(FROZEN_SOIL=1, frozen_compat=0, Nnode=int(rng.choice([5, 10, 12, 21, 24])), NOFLUX=int(rng.integers(0, 2)),
                      EXP_TRANS=int(rng.integers(0, 2)))
            mode = rng.integers(0, 4)
            if mode == 1 and kw["Nnode"] <= 21:
                kw["IMPLICIT"] = 1
            elif mode >= 2:
                kw["QUICK_SOLVE"] = 1
        elif rng.integers(0, 3) == 0:
            kw = dict(FULL_ENERGY=0, dt=int(rng.choice([3, 24])), snow_step=3, BLOWING=1)
        kw["Nband"] = int(rng.integers(1, 4))
        kw["GRND_FLUX_TYPE"] = int(rng.choice([C["VIC_GF_406"], C["VIC_GF_410"], C["VIC_GF_FULL"]]))
        kw["AERO_RESIST_CANSNOW"] = int(rng.choice([C["VIC_AR_406"], C["VIC_AR_406_LS"], C["VIC_AR_406_FULL"], C["VIC_AR_410"], C["VIC_AR_COMBO"]]))
        kw["SNOW_DENSITY"] = int(rng.integers(0, 2)); kw["SNOW_ALBEDO"] = int(rng.integers(0, 2)); kw["TEMP_TH_TYPE"] = int(rng.integers(0, 2))
        kw["CORRPREC"] = int(rng.integers(0, 2)); kw["TFALLBACK"] = int(rng.integers(0, 4) > 0)
        glacier = bool(rng.integers(0, 3) == 0) and kw["Nband"] > 1
        if glacier:
            kw["GLACIER_DYNAMICS"] = int(rng.integers(0, 2))
        out.append(("combo3_%02d" % k, dict(kw=kw, variant="fixed" if kw.get("FROZEN_SOIL") else "plain", ncell=5, ntile=2, glacier=glacier,
                                            nsteps=24, doy=int(rng.choice([20, 95, 330, 350])), tweak="windy" if kw.get("BLOWING") else None)))
    return out


RANDOM_COMBINATIONS = dict(random_combinations())
RANDOM_COMBINATIONS.update(random_combinations_round3())


def all_scenarios():
    """Every scenario name, ordered for a single test process: the IMPLICIT ones last and by increasing number of unknowns
    (see IMPLICIT_BRANCHES: left-overs in the patched reference's static work arrays)."""
    every = dict(OPTION_BRANCHES); every.update(QUICK_SOLVE_BRANCHES); every.update(RANDOM_COMBINATIONS); every.update(IMPLICIT_BRANCHES)
    plain = [n for n, sp in every.items() if not sp["kw"].get("IMPLICIT")]
    imp = sorted([n for n, sp in every.items() if sp["kw"].get("IMPLICIT")],
                 key=lambda n: every[n]["kw"].get("Nnode", 10) - (1 if every[n]["kw"].get("NOFLUX") else 2))
    return plain + imp, every
