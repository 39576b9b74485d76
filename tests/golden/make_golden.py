"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libvicref*.so, built from /root/reference by
oracle/ref_build/build_ref.sh).  Run in the build container only:  python tests/golden/make_golden.py

Each fixture is data only: the domain tables, the forcing, the state after initialize_model_state, and the reference's
state / fluxes / cell outputs after every `stride`-th step of a free run.  Nothing from the reference's sources is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vic_amd import abi, domain  # noqa: E402
from vic_amd.abi import C  # noqa: E402
from oracle.pyref import RefModel  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SCENARIOS = {
    # name: (option kwargs, variant, ncell, ntile, glacier, nsteps, start_doy, stride)
    "quickflux_winter": (dict(FULL_ENERGY=1), "plain", 6, 3, False, 96, 10, 8),
    "quickflux_melt": (dict(FULL_ENERGY=1), "plain", 6, 3, False, 120, 85, 8),
    "quickflux_summer": (dict(FULL_ENERGY=1), "plain", 6, 3, False, 96, 190, 8),
    "bands_glacier": (dict(FULL_ENERGY=1, Nband=3), "plain", 4, 2, True, 120, 150, 8),
    "waterbalance_daily": (dict(FULL_ENERGY=0, dt=24, snow_step=3), "plain", 6, 2, False, 120, 300, 10),
    "frozen_fixed": (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, frozen_compat=0), "fixed", 4, 3, False, 72, 20, 6),
    "frozen_compat": (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, frozen_compat=1), "compat", 4, 3, False, 48, 20, 6),
    "frozen_fixed_glacier": (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=8, Nband=2, frozen_compat=0), "fixed", 4, 2, True, 72, 110, 6),
    # the bench workloads' shape (bench.config cfg3 / cfg4: 5 bands x 5 tiles, 10 nodes, "fixed", start_doy 60), without and
    # with the glacier slot in the top band
    "frozen_cfg3_shape": (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=0), "fixed", 4, 5, False, 48, 60, 6),
    "frozen_cfg4_shape_glacier": (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5, frozen_compat=0), "fixed", 4, 5, True, 48, 60, 6),
}


def make_pure():
    """Known-answer vectors of the pure functions (tests/pure_inputs.py) from the reference's own functions."""
    from tests.pure_inputs import pure_inputs, OPTION_SETS
    inputs = pure_inputs()
    for oname, kw in OPTION_SETS.items():
        opt = abi.default_options(**kw)
        d = domain.make_domain(2, opt, ntile=1)
        ref = RefModel(d, "plain")
        out = {}
        for fn, inp in inputs.items():
            out["in_%d" % fn] = inp
            out["out_%d" % fn] = ref.pure(fn, inp)
        ref.close()
        np.savez_compressed(os.path.join(HERE, "pure_%s.npz" % oname), **out)
        print("pure_%s" % oname, "functions", len(inputs))


def make_gmb():
    """Glacier mass-balance fit (accumulateGlacierMassBalance.c:53-66): the reference's state before the fit and its fit."""
    from tests.test_oracle import GMB_CASES, _gmb_setup
    for name, kw, glacier in GMB_CASES:
        d, f, sf, dmy = _gmb_setup(kw, glacier)
        ref = RefModel(d, "plain")
        ref.init_state(f[0], dmy[0], d.init_moist)
        sd0, si0 = ref.get_state()
        isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
        sd0[C["SD_GLAC_CUM_MASS_BALANCE"], isg] = 0.0
        ref.set_state(sd0, si0)
        for s in range(f.shape[0]):
            ref.step(f[s], sf[s], dmy[s])
        sd, si = ref.get_state()
        eq = ref.glacier_fit(reset=True)
        ref.close()
        np.savez_compressed(os.path.join(HERE, "gmb_%s.npz" % name), sd=sd, si=si, eq=eq, cell_params=d.cell_params,
                            hru_iparams=d.hru_iparams)
        print("gmb_%s" % name, eq[:, 0])


def make_forcing_derive():
    """atmos[rec] as the reference's initialize_atmos derives it from in-memory forcing records (tests/test_forcing_stream.py)."""
    from tests.test_forcing_stream import DERIVE_CASES, derive_case
    for name in sorted(DERIVE_CASES):
        d, file, raw, force_dt, min_wind, plapse = derive_case(name)
        ref = RefModel(d, "plain")
        f, sf = ref.derive_forcing(file, force_dt, min_wind, plapse)
        ref.close()
        np.savez_compressed(os.path.join(HERE, "forcing_derive_%s.npz" % name), records=file, forcing=f, snowflag=sf)
        print("forcing_derive_%s" % name, f.shape)


def main():
    only = sys.argv[1:]                   # optional: the trajectory fixtures to (re)generate, by name
    if only == ["forcing_derive"]:
        return make_forcing_derive()
    if not only:
        make_pure()
        make_gmb()
        make_forcing_derive()
    for name, (kw, variant, ncell, ntile, glacier, nsteps, doy, stride) in SCENARIOS.items():
        if only and name not in only:
            continue
        opt = abi.default_options(**kw)
        d = domain.make_domain(ncell, opt, ntile=ntile, glacier_top_band=glacier)
        f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=doy)
        ref = RefModel(d, variant)
        ref.init_state(f[0], dmy[0], d.init_moist)
        cp = ref.get_cell_params()
        assert np.array_equal(cp, d.cell_params, equal_nan=True)
        sd0, si0 = ref.get_state()
        if glacier:
            isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
            sd0[C["SD_GLAC_CUM_MASS_BALANCE"], isg] = 0.0
            ref.set_state(sd0, si0)
        states_d, states_i, fluxes, cells, steps = [], [], [], [], []
        for s in range(nsteps):
            fx, co, ce = ref.step(f[s], sf[s], dmy[s])
            assert ce.sum() == 0
            if (s + 1) % stride == 0:
                sd, si = ref.get_state()
                states_d.append(sd); states_i.append(si); fluxes.append(fx); cells.append(co); steps.append(s)
        ref.close()
        optv = np.array([getattr(opt, fld) for fld, _ in abi.Options._fields_ if not fld.startswith("reserved")], dtype=np.float64)
        optn = np.array([fld for fld, _ in abi.Options._fields_ if not fld.startswith("reserved")])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), opt_names=optn, opt_values=optv, veglib=d.veglib,
                            cell_params=d.cell_params, hru_iparams=d.hru_iparams, hru_dparams=d.hru_dparams,
                            cell_hru_offset=d.cell_hru_offset, cell_hru_list=d.cell_hru_list, init_moist=d.init_moist,
                            forcing=f, snowflag=sf, dmy=dmy, sd0=sd0, si0=si0, steps=np.array(steps),
                            states_d=np.stack(states_d), states_i=np.stack(states_i), fluxes=np.stack(fluxes),
                            cells=np.stack(cells), variant=np.array(variant))
        print(name, "ok", os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
