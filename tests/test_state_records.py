"""The state in the order of the reference's state file (write_model_state.c:95-337, include/vicgpu.h SR_*).

  oracle vs reference   the records vicorc_state_records writes, laid end to end per cell, ARE the per-HRU part of the stream
                        the reference's own processCellForStateFile produces (through an in-memory StateIO back-end in the
                        shim), value for value; and reading that stream back through the reference's reader into a fresh
                        model changes exactly the rows the records scatter, to exactly the same values.
  device vs oracle      vicgpu_get_state_records == the oracle's records of the same state, bit for bit; the scatter lands on
                        the same state; a run restarted from records continues like the oracle restarted from them.
"""
import numpy as np
import pytest

from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from tests.util import rel_diff, worst

FROZEN = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, frozen_compat=0)
CASES = [
    ("quickflux_bands", dict(FULL_ENERGY=1, Nband=3), "plain", 6, 3, False, 40, 70),
    ("frozen_glacier", dict(FROZEN, Nband=2), "fixed", 4, 2, True, 30, 110),
    ("wb_daily_bare", dict(FULL_ENERGY=0, dt=24, snow_step=3), "plain", 6, 3, False, 20, 330),
]


def _setup(kw, ncell, ntile, glacier, nsteps, doy, bare=False):
    opt = abi.default_options(**kw)
    d = domain.make_domain(ncell, opt, ntile=ntile, glacier_top_band=glacier, **(dict(bare_fraction=0.3) if bare else {}))
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=doy)
    return d, f, sf, dmy


def records_to_stream(d, rec):
    """The per-HRU part of the stream: every record end to end, without the Wdew slot of artificial bare-soil HRUs
    (write_model_state.c:240-242)."""
    out = []
    bare = d.hru_iparams[C["HPI_IS_ARTIFICIAL_BARE"]]
    for k, g in enumerate(d.cell_hru_list):
        r = rec[k]
        out.append(np.delete(r, C["SR_WDEW"]) if bare[g] else r)
    return out


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_records_are_the_reference_stream(case, oracle_lib, ref_available):
    if not ref_available:
        pytest.skip("reference build (oracle/_ref) not available")
    name, kw, variant, ncell, ntile, glacier, nsteps, doy = case
    d, f, sf, dmy = _setup(kw, ncell, ntile, glacier, nsteps, doy, bare=name.endswith("bare"))
    Nn = d.opt.Nnode
    ref = oracle_lib.RefModel(d, variant)
    ref.init_state(f[0], dmy[0], d.init_moist)
    sd0, si0 = ref.get_state()
    orc = oracle_lib.OracleModel(d)
    orc.set_state(sd0, si0); orc.set_fluxes(ref.get_fluxes())
    for s in range(nsteps):
        ref.step(f[s], sf[s], dmy[s]); orc.step(f[s], sf[s], dmy[s])
    vals, ids, start = ref.state_stream()
    rec = orc.get_state_records()
    per_hru = records_to_stream(d, rec)
    head = 2 * Nn + 4                      # dz_node, Zsum_node, the 4 glacier mass balance equation terms (write_model_state.c:120-160)
    k = 0
    for c in range(d.ncell):
        cell_vals = vals[start[c] + head:start[c + 1]]
        mine = np.concatenate([per_hru[k + i] for i in range(d.cell_hru_offset[c + 1] - d.cell_hru_offset[c])])
        k += d.cell_hru_offset[c + 1] - d.cell_hru_offset[c]
        assert cell_vals.shape == mine.shape, "cell %d: the reference streams %d values, the records hold %d" % (c, len(cell_vals), len(mine))
        assert np.array_equal(cell_vals, mine, equal_nan=True), "cell %d: first difference at value %d" % (
            c, int(np.flatnonzero(~((cell_vals == mine) | (np.isnan(cell_vals) & np.isnan(mine))))[0]))
        assert ids[start[c] + head] == ref.state_var_id("HRU_BAND_INDEX")
    # the read side: a fresh model on both sides, the reference reads its own stream, the oracle scatters the records
    ref2 = oracle_lib.RefModel(d, variant)
    ref2.init_state(f[0], dmy[0], d.init_moist)
    orc2 = oracle_lib.OracleModel(d)
    orc2.set_state(*ref2.get_state()); orc2.set_fluxes(ref2.get_fluxes())
    assert ref2.read_state_stream(vals, ids) == 0
    assert orc2.set_state_records(rec) == 0
    (sr, ir), (so, io) = ref2.get_state(), orc2.get_state()
    assert np.array_equal(sr, so, equal_nan=True), worst(sr, so, "SD_", 1e-300)[1]
    assert np.array_equal(ir, io)
    fr, fo = ref2.get_fluxes(), orc2.get_fluxes()
    for row in ("FX_GLAC_VAPOR_FLUX", "FX_SNOW_SURFACE_FLUX", "FX_SNOW_VAPOR_FLUX"):
        assert np.array_equal(fr[C[row]], fo[C[row]], equal_nan=True), row
    # a record of the wrong band is refused (write_model_state.c:179-188) and nothing is read
    bad = rec.copy(); bad[1, C["SR_BAND_INDEX"]] += 1
    before = orc2.get_state()[0].copy()
    assert orc2.set_state_records(bad) == 2
    assert np.array_equal(before, orc2.get_state()[0], equal_nan=True)
    ref.close(); ref2.close()


GPU_CASES = [
    ("quickflux_bands", dict(FULL_ENERGY=1, Nband=3), 40, 3, False, 24, 70),
    ("frozen_glacier", dict(FROZEN, Nband=2), 10, 2, True, 16, 110),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_device_state_records(case, oracle_lib):
    from vic_amd.api import Model, VicGpuError
    name, kw, ncell, ntile, glacier, nsteps, doy = case
    d, f, sf, dmy = _setup(kw, ncell, ntile, glacier, 2 * nsteps, doy)
    sd0, si0 = init_state.initial_state(d, f[0])
    if glacier:
        sd0[C["SD_GLAC_CUM_MASS_BALANCE"], d.hru_iparams[C["HPI_IS_GLACIER"]] != 0] = 0.0
    orc = oracle_lib.OracleModel(d); orc.set_state(sd0, si0)
    for s in range(nsteps):
        orc.step(f[s], sf[s], dmy[s])
    sd, si = orc.get_state()
    fx = orc.get_fluxes()
    gpu = Model(d); gpu.set_state(sd, si); gpu.set_fluxes(fx)
    rec_o, rec_g = orc.get_state_records(), gpu.get_state_records()
    assert np.array_equal(rec_o, rec_g, equal_nan=True), "gather: %d values differ" % (~((rec_o == rec_g) | (np.isnan(rec_o) & np.isnan(rec_g)))).sum()
    # restart: fresh contexts holding the initial state, the records scattered into them
    orc2 = oracle_lib.OracleModel(d); orc2.set_state(sd0, si0)
    gpu2 = Model(d); gpu2.set_state(sd0, si0)
    assert orc2.set_state_records(rec_o) == 0
    gpu2.set_state_records(rec_g)
    (so, io), (sg, ig) = orc2.get_state(), gpu2.get_state()
    assert np.array_equal(so, sg, equal_nan=True) and np.array_equal(io, ig)
    rows = [C[r] for r in ("FX_GLAC_VAPOR_FLUX", "FX_SNOW_SURFACE_FLUX", "FX_SNOW_VAPOR_FLUX")]
    assert np.array_equal(orc2.get_fluxes()[rows], gpu2.get_fluxes()[rows], equal_nan=True)
    # ... and both continue alike (teacher-forced on the oracle's trajectory, the bound of tests/test_gpu_parity.py)
    gpu2.push_forcing(f, sf, dmy)
    for s in range(nsteps, 2 * nsteps):
        sd_in, si_in = orc2.get_state()
        orc2.step(f[s], sf[s], dmy[s])
        gpu2.set_state(sd_in, si_in); gpu2.dist_prec(s, 1)
        so, sg = orc2.get_state()[0], gpu2.get_state()[0]
        so[C["SD_ERROR"]] = 0; sg[C["SD_ERROR"]] = 0
        w, msg = worst(so, sg, "SD_", floor=1e-6)
        assert w < 1e-6, "step %d after the restart: %s" % (s, msg)
    bad = rec_g.copy(); bad[3, C["SR_VEG_CLASS"]] += 1
    with pytest.raises(VicGpuError):
        gpu2.set_state_records(bad)
    # interrupted == uninterrupted, bit for bit: device A runs 2k steps; device B runs k, hands its records and its tables to
    # a new context (tables first, the records scattered on top of them), which runs the other k
    a = Model(d); a.set_state(sd0, si0); a.push_forcing(f, sf, dmy); a.dist_prec(0, 2 * nsteps)
    b = Model(d); b.set_state(sd0, si0); b.push_forcing(f, sf, dmy); b.dist_prec(0, nsteps)
    rec_b, (sd_b, si_b), fx_b = b.get_state_records(), b.get_state(), b.get_fluxes()
    c2 = Model(d); c2.set_state(sd_b, si_b); c2.set_fluxes(fx_b); c2.set_state_records(rec_b)
    c2.push_forcing(f, sf, dmy); c2.dist_prec(nsteps, nsteps)
    (sa, ia), (sc, ic) = a.get_state(), c2.get_state()
    assert np.array_equal(sa, sc, equal_nan=True) and np.array_equal(ia, ic)
