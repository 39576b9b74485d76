"""Forcing streaming (vicgpu_prefetch_forcing / _raw / vicgpu_swap_forcing, include/vicgpu.h) and the on-device derivation of
atmos[rec] from hourly raw forcing (initialize_atmos.c:290-295, 518-536, 886-893, 980-1000, 1175-1193, 1275-1303)."""
import numpy as np
import pytest

from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from tests.util import rel_diff, worst


def raw_from_table(d, f):
    """Hourly raw forcing whose derivation gives back the sub-step values of a derived table (snow_step == 1 hour)."""
    opt = d.opt
    assert opt.snow_step == 1
    n = f.shape[0]
    raw = np.zeros((n, C["VIC_NRAW"], opt.dt, d.ncell))
    for name, src, scale in (("VIC_RAW_AIR_TEMP", "VIC_F_AIR_TEMP", 1.0), ("VIC_RAW_PREC", "VIC_F_PREC", 1.0),
                             ("VIC_RAW_PRESSURE_KPA", "VIC_F_PRESSURE", 1e-3), ("VIC_RAW_VP_KPA", "VIC_F_VP", 1e-3),
                             ("VIC_RAW_SHORTWAVE", "VIC_F_SHORTWAVE", 1.0), ("VIC_RAW_LONGWAVE", "VIC_F_LONGWAVE", 1.0),
                             ("VIC_RAW_WIND", "VIC_F_WIND", 1.0)):
        raw[:, C[name]] = f[:, C[src], :opt.NF] * scale
    return raw


@pytest.mark.parametrize("kw", [dict(FULL_ENERGY=1, Nband=3), dict(FULL_ENERGY=1, dt=3, snow_step=1, Nband=2, TEMP_TH_TYPE=0)],
                         ids=["hourly", "3hourly_substeps_vic412"])
def test_oracle_derivation_reproduces_the_synthetic_table(kw, oracle_lib):
    """domain.make_forcing writes atmos[rec] the way initialize_atmos leaves it (Pa, vpd, PLAPSE density, snowflag of the
    coldest band); fed the same sub-step values as hourly raw input, the derivation must give that table back."""
    opt = abi.default_options(**kw)
    d = domain.make_domain(40, opt, ntile=2)
    f, sf, dmy = domain.make_forcing(d, 0, 30, start_doy=80)
    orc = oracle_lib.OracleModel(d)
    f2, sf2 = orc.derive_forcing(raw_from_table(d, f), min_wind=0.0, plapse=1)
    assert rel_diff(f, f2, 1e-12).max() < 1e-13, worst(f.reshape(-1, d.ncell), f2.reshape(-1, d.ncell), "VIC_F_", 1e-12)[1]
    assert np.array_equal(sf, sf2)
    # the floor on the wind speed and the clipped vapour pressure deficit
    raw = raw_from_table(d, f)
    raw[:, C["VIC_RAW_WIND"]] *= 0.05
    raw[::2, C["VIC_RAW_VP_KPA"]] *= 3.0
    f3, _ = orc.derive_forcing(raw, min_wind=0.1, plapse=0)
    assert f3[:, C["VIC_F_WIND"]].min() >= 0.1 and (f3[:, C["VIC_F_VPD"]] == 0).any() and f3[:, C["VIC_F_VPD"]].min() >= 0
    T, pr = f3[:, C["VIC_F_AIR_TEMP"], 0], f3[:, C["VIC_F_PRESSURE"], 0]
    assert rel_diff(f3[:, C["VIC_F_DENSITY"], 0], 0.003486 * pr / (275.0 + T), 1e-12).max() < 1e-14


# (options, forcing-file time step, min wind, PLAPSE): hourly forcing with one sub-step per step (the bench configs), with three
# sub-steps per step (sub-index NR), 3-hourly records for a daily step with 3-hourly snow steps, and the non-PLAPSE density
DERIVE_CASES = {
    "hourly": (dict(FULL_ENERGY=1, Nband=3), 1, 0.3, 1),
    "3hourly_step_hourly_substeps_vic412": (dict(FULL_ENERGY=1, dt=3, snow_step=1, Nband=2, TEMP_TH_TYPE=0), 1, 0.3, 1),
    "daily_step_3hourly_records": (dict(FULL_ENERGY=0, dt=24, snow_step=3), 3, 0.25, 1),
    "hourly_no_plapse": (dict(FULL_ENERGY=1, Nband=3), 1, 0.0, 0),
}


def derive_case(name, ncell=7, ndays=3):
    """(domain, file records [VIC_NRAW][nfile][ncell] in file units, hourly raw table as vicgpu_prefetch_forcing_raw takes it)."""
    kw, force_dt, min_wind, plapse = DERIVE_CASES[name]
    opt = abi.default_options(**kw)
    d = domain.make_domain(ncell, opt, ntile=2)
    nfile, nsteps = ndays * 24 // force_dt, ndays * 24 // opt.dt
    rng = np.random.default_rng(sorted(DERIVE_CASES).index(name) + 11)
    hrs = (np.arange(nfile) * force_dt)[:, None]
    file = np.zeros((C["VIC_NRAW"], nfile, d.ncell))
    file[C["VIC_RAW_AIR_TEMP"]] = -2 + 6 * np.sin(2 * np.pi * (hrs - 9) / 24) + rng.normal(0, 1, (nfile, d.ncell))
    file[C["VIC_RAW_PREC"]] = np.where(rng.uniform(size=(nfile, d.ncell)) < 0.3, rng.uniform(0, 3, (nfile, d.ncell)), 0.0)
    file[C["VIC_RAW_PRESSURE_KPA"]] = 85 + rng.normal(0, 0.5, (nfile, d.ncell))
    file[C["VIC_RAW_VP_KPA"]] = 0.45 + rng.uniform(-0.2, 0.4, (nfile, d.ncell))          # partly above saturation: vpd clips
    file[C["VIC_RAW_SHORTWAVE"]] = np.maximum(0, 400 * np.sin(2 * np.pi * (hrs - 6) / 24)) * rng.uniform(0.5, 1, (nfile, d.ncell))
    file[C["VIC_RAW_LONGWAVE"]] = 250 + rng.normal(0, 10, (nfile, d.ncell))
    file[C["VIC_RAW_WIND"]] = rng.uniform(0.0, 4.0, (nfile, d.ncell))                     # partly below the floor
    # the hourly table: what initialize_atmos makes of the records before it aggregates them (local_forcing_data,
    # initialize_atmos.c:352-392): a record's value in each of its hours, amounts divided by the record length
    hourly = np.repeat(file, force_dt, axis=1)
    hourly[C["VIC_RAW_PREC"]] = np.repeat(file[C["VIC_RAW_PREC"]] / force_dt, force_dt, axis=0)
    raw = np.ascontiguousarray(hourly.reshape(C["VIC_NRAW"], nsteps, opt.dt, d.ncell).transpose(1, 0, 2, 3))
    return d, file, raw, force_dt, min_wind, plapse


@pytest.mark.parametrize("name", sorted(DERIVE_CASES))
def test_oracle_derivation_equals_reference_initialize_atmos(name, oracle_lib):
    """The reference's own initialize_atmos (initialize_atmos.c:7-1349, MTCLIM included) run on forcing records held in
    memory -- the harness only plays read_atmos_data -- against the oracle's derivation from the hourly table: every value of
    atmos[rec] and every snowflag, bit for bit."""
    if not oracle_lib.have_ref("plain"):
        pytest.skip("reference build not present (oracle/_ref)")
    d, file, raw, force_dt, min_wind, plapse = derive_case(name)
    ref = oracle_lib.RefModel(d, "plain")
    fr, sr = ref.derive_forcing(file, force_dt, min_wind, plapse)
    ref.close()
    fo, so = oracle_lib.OracleModel(d).derive_forcing(raw, min_wind, plapse)
    assert np.array_equal(sr, so)
    assert np.array_equal(fr, fo), worst(fr.reshape(-1, d.ncell), fo.reshape(-1, d.ncell), "VIC_F_", 1e-300)[1]
    assert (fr[:, C["VIC_F_VPD"]] == 0).any() and sr.any()


@pytest.mark.parametrize("name", sorted(DERIVE_CASES))
def test_oracle_derivation_equals_reference_fixture(name, oracle_lib):
    """The same comparison against the committed outputs of the reference (tests/golden/forcing_derive_*.npz, written by
    tests/golden/make_golden.py from the run above): what travels to the GPU box."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "forcing_derive_%s.npz" % name))
    d, file, raw, force_dt, min_wind, plapse = derive_case(name)
    assert np.array_equal(z["records"], file)
    fo, so = oracle_lib.OracleModel(d).derive_forcing(raw, min_wind, plapse)
    assert np.array_equal(z["snowflag"], so) and np.array_equal(z["forcing"], fo)


def _state0(d, f):
    return init_state.initial_state(d, f[0])


@pytest.mark.gpu
@pytest.mark.parametrize("kw,ncell", [(dict(FULL_ENERGY=1, Nband=3), 300), (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=2, frozen_compat=0), 24)],
                         ids=["quickflux", "frozen"])
def test_streamed_chunks_equal_one_chunk(kw, ncell):
    """48 steps as one resident chunk == 4 chunks of 12 with the next chunk uploading while the current one runs (one of them
    from pinned memory, the others through the staging area): same bits."""
    from vic_amd.api import Model
    opt = abi.default_options(**kw)
    d = domain.make_domain(ncell, opt, ntile=2)
    nsteps, nchunk = 48, 4
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=75)
    sd0, si0 = _state0(d, f)
    a = Model(d); a.set_state(sd0, si0); a.push_forcing(f, sf, dmy); a.dist_prec(0, nsteps)
    want = a.get_state(); want_acc = a.get_accum()
    b = Model(d); b.set_state(sd0, si0)
    n = nsteps // nchunk
    pf, psf = b.pinned((n,) + f.shape[1:]), b.pinned((n,) + sf.shape[1:], np.uint8)
    b.push_forcing(f[:n], sf[:n], dmy[:n])
    for k in range(nchunk):
        if k + 1 < nchunk:
            lo, hi = (k + 1) * n, (k + 2) * n
            if k == 1:
                pf[...] = f[lo:hi]; psf[...] = sf[lo:hi]
                b.prefetch_forcing(pf, psf, dmy[lo:hi])
            else:
                b.prefetch_forcing(f[lo:hi].copy(), sf[lo:hi].copy(), dmy[lo:hi])
        b.dist_prec(0, n, sync=False)
        if k + 1 < nchunk:
            b.swap_forcing()
    got = b.get_state()
    assert np.array_equal(want[0], got[0], equal_nan=True) and np.array_equal(want[1], got[1])
    assert np.array_equal(want_acc, b.get_accum(), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(FULL_ENERGY=1, Nband=3), dict(FULL_ENERGY=0, dt=24, snow_step=3), dict(FULL_ENERGY=1, dt=3, snow_step=1, TEMP_TH_TYPE=0)],
                         ids=["hourly", "daily_3h_substeps", "3hourly_vic412"])
def test_device_derivation_against_oracle(kw, oracle_lib):
    from vic_amd.api import Model
    opt = abi.default_options(**kw)
    d = domain.make_domain(70, opt, ntile=2)
    nsteps = 12
    rng = np.random.default_rng(3)
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=80)
    raw = np.zeros((nsteps, C["VIC_NRAW"], opt.dt, d.ncell))
    hours = lambda v: np.repeat(f[:, C[v], :opt.NF], opt.snow_step, axis=1)          # the sub-step value in each of its hours
    raw[:, C["VIC_RAW_AIR_TEMP"]] = hours("VIC_F_AIR_TEMP") + rng.normal(0, 1.0, raw[:, 0].shape)
    raw[:, C["VIC_RAW_PREC"]] = hours("VIC_F_PREC") / opt.snow_step
    raw[:, C["VIC_RAW_PRESSURE_KPA"]] = hours("VIC_F_PRESSURE") * 1e-3
    raw[:, C["VIC_RAW_VP_KPA"]] = hours("VIC_F_VP") * 1e-3 * rng.uniform(0.5, 1.6, raw[:, 0].shape)      # some above saturation
    raw[:, C["VIC_RAW_SHORTWAVE"]] = hours("VIC_F_SHORTWAVE"); raw[:, C["VIC_RAW_LONGWAVE"]] = hours("VIC_F_LONGWAVE")
    raw[:, C["VIC_RAW_WIND"]] = hours("VIC_F_WIND") * rng.uniform(0.0, 1.2, raw[:, 0].shape)
    orc = oracle_lib.OracleModel(d)
    fo, so = orc.derive_forcing(raw, min_wind=0.4, plapse=1)
    gpu = Model(d)
    gpu.prefetch_forcing_raw(raw, dmy, min_wind_speed=0.4, plapse=True); gpu.swap_forcing()
    dev_f = []
    for s in range(nsteps):
        fg, sg = gpu.get_forcing(s)
        dev_f.append(fg)
        assert np.array_equal(so[s], sg), "step %d snowflag" % s
        rows = [C[v] for v in ("VIC_F_AIR_TEMP", "VIC_F_PREC", "VIC_F_PRESSURE", "VIC_F_DENSITY", "VIC_F_SHORTWAVE", "VIC_F_LONGWAVE", "VIC_F_WIND")]
        assert np.array_equal(fo[s][rows], fg[rows]), "step %d" % s                  # no transcendental on these rows: bit for bit
        assert rel_diff(fo[s], fg, 1e-6).max() < 1e-11                               # vp / vpd = svp(T) - vp go through exp: a few ulp of svp
    assert (fo[:, C["VIC_F_VPD"]] == 0).any() and (raw[:, C["VIC_RAW_WIND"]] < 0.4).any()
    # and the path runs on it.  The oracle steps on the table the DEVICE derived (read back): vp / vpd differ from the oracle's
    # own derivation in the last bits of exp(), and saturated air (vp == svp(T), vpd == 0) sits on branch points of the
    # canopy energy balance, where last-bit differences of the inputs pick different branches
    # With snow sub-steps inside a step (NF > 1) an HRU whose intercepted canopy snow melts out during the step can do so one
    # sub-step earlier or later on the two sides (the last bits of the device's exp() in svp decide a sign at 0 C; the -O1 and
    # -O3 device builds agree with each other to the last digit, and the host build of the same sources agrees with the oracle):
    # such an HRU-step is a different -- equally valid -- branch, not an error of the derivation.  At most 2 of the 1680
    # HRU-steps of this run may differ, and only HRUs that enter the step with snow in the canopy.
    sd0, si0 = _state0(d, fo)
    orc.set_state(sd0, si0); gpu.set_state(sd0, si0)
    outliers = 0
    for s in range(nsteps):
        sd_in, si_in = orc.get_state()
        orc.step(dev_f[s], so[s], dmy[s])
        gpu.set_state(sd_in, si_in); gpu.dist_prec(s, 1)
        a, b = orc.get_state()[0], gpu.get_state()[0]
        a[C["SD_ERROR"]] = 0; b[C["SD_ERROR"]] = 0
        bad = rel_diff(a, b, 1e-6).max(axis=0) >= 1e-6
        if opt.NF > 1 and bad.any():
            assert (sd_in[C["SD_SNOW_CANOPY"], bad] > 0).all(), "step %d %s" % (s, worst(a, b, "SD_", floor=1e-6)[1])
            outliers += int(bad.sum())
            a, b = a[:, ~bad], b[:, ~bad]
        w, msg = worst(a, b, "SD_", floor=1e-6)
        assert w < 1e-6, "step %d %s" % (s, msg)
    assert outliers <= 2, outliers
