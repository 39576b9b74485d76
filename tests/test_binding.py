"""The reference-side binding (integration/vicgpu_binding.{h,cpp}): C++ against the reference's own headers, compiled by
oracle/ref_build/build_ref.sh and linked into the reference harness.

  not gpu   what it packs from the reference's structs (veg_lib_struct[], soil_con_struct, hruList, HRU state, ProgramState)
            is exactly the tables those structs were built from -- every row, through its own HRU numbering;
  gpu       the reference run twice on the same cells: once by its own full_energy, once with VicGpuBinding in place of the
            cell loop of vicNl.c:506-593 (reference structs -> libvicgpu.so -> reference structs).
Needs oracle/_ref (built where /root/reference is present; the prebuilt libraries travel to the GPU box)."""
import numpy as np
import pytest

from vic_amd import abi, domain
from vic_amd.abi import C
from tests.util import edge_domain, worst

FROZEN = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=2)


def _perm(d, t):
    """binding HRU id of every harness HRU id (both number the same (cell, position in hruList) pairs)."""
    p = np.zeros(d.nhru, dtype=np.int64)
    for c in range(d.ncell):
        a, b = d.cell_hru_offset[c], d.cell_hru_offset[c + 1]
        mine = t["cell_list"][t["cell_off"][c]:t["cell_off"][c + 1]]
        assert len(mine) == b - a
        p[d.cell_hru_list[a:b]] = mine
    return p


@pytest.mark.parametrize("kind", ["regular_glacier", "irregular"])
def test_binding_packs_the_tables_the_structs_were_built_from(kind, oracle_lib, ref_available):
    if not ref_available:
        pytest.skip("reference build (oracle/_ref) not available")
    opt = abi.default_options(**dict(FROZEN, frozen_compat=0, TFALLBACK=1, GRND_FLUX_TYPE=C["VIC_GF_FULL"], CORRPREC=1))
    d = domain.make_domain(6, opt, ntile=3, glacier_top_band=True) if kind == "regular_glacier" else edge_domain(opt, ncell=12)
    f, sf, dmy = domain.make_forcing(d, 0, 2, start_doy=40)
    ref = oracle_lib.RefModel(d, "fixed")
    ref.init_state(f[0], dmy[0], d.init_moist)
    t = ref.binding_tables()
    o, bo = d.opt, t["opt"]
    for name, _ in abi.Options._fields_:
        if name in ("frozen_compat", "NODE_SOLVER") or name.startswith("reserved"):
            continue                                   # not options of the reference: constructor arguments of the binding
        assert getattr(o, name) == getattr(bo, name), name
    # veg library: RGL is a float, overstory a flag in the reference
    want = d.veglib.copy(); want[:, C["VL_RGL"]] = np.float32(want[:, C["VL_RGL"]]); want[:, C["VL_OVERSTORY"]] = want[:, C["VL_OVERSTORY"]] != 0
    assert np.array_equal(want, t["veglib"])
    # cell table: elevation, lat and the band elevations are floats in soil_con_struct; the node rows are what
    # initialize_model_state wrote (vicref_get_cell_params reads them the same way)
    want = ref_cell_params = np.zeros_like(t["cell_params"])
    fn = ref.lib.vicref_get_cell_params; fn.restype = int
    import ctypes
    fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    assert fn(ref.h, want.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    for r in [C["CP_ELEVATION"], C["CP_LAT"]] + [abi.cp_band(C["CPB_BANDELEV"], b, o.Nnode, o.Nband) for b in range(o.Nband)]:
        want[r] = np.float32(want[r])
    w, msg = worst(want, t["cell_params"], "CP_", 1e-300)
    assert np.array_equal(want, t["cell_params"]), msg
    # HRU tables and state through the two numberings
    p = _perm(d, t)
    hpd = d.hru_dparams.copy()
    for l in range(3):
        hpd[C["HPD_ROOT0"] + l] = np.float32(hpd[C["HPD_ROOT0"] + l])
    assert np.array_equal(d.hru_iparams, t["hpi"][:, p]) and np.array_equal(hpd, t["hpd"][:, p])
    sd, si = ref.get_state()
    assert np.array_equal(sd, t["sd"][:, p], equal_nan=True) and np.array_equal(si, t["si"][:, p])
    # position-major numbering: the first HRUs of all cells come first
    first = t["cell_list"][t["cell_off"][:-1][np.diff(t["cell_off"]) > 0]]
    assert np.array_equal(np.sort(first), np.arange(len(first)))
    ref.close()


GPU_CASES = [
    ("quickflux_bands", dict(FULL_ENERGY=1, Nband=3), "plain", False, 24, 70),
    ("frozen_fixed", dict(FROZEN, frozen_compat=0), "fixed", False, 12, 330),
    ("frozen_compat_glacier", dict(FROZEN, frozen_compat=1), "compat", True, 12, 100),
    ("wb_daily", dict(FULL_ENERGY=0, dt=24, snow_step=3), "plain", False, 10, 60),
    # BLOWING: the binding hands veg_con's sigma_slope / lag_one / fetch to the device (HPD rows of ABI v3); strong wind
    ("blowing_bands", dict(FULL_ENERGY=1, Nband=2, BLOWING=1), "plain", False, 24, 350),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_reference_runs_through_the_binding(case, oracle_lib, ref_available):
    if not ref_available:
        pytest.skip("reference build (oracle/_ref) not available")
    name, kw, variant, glacier, nsteps, doy = case
    opt = abi.default_options(**kw)
    d = domain.make_domain(40, opt, ntile=2, glacier_top_band=glacier)
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=doy)
    if name.startswith("blowing"):
        f[:, C["VIC_F_WIND"]] *= 3.5
    a, b = oracle_lib.RefModel(d, variant), oracle_lib.RefModel(d, variant)
    for m in (a, b):
        m.init_state(f[0], dmy[0], d.init_moist)
        sd0, si0 = m.get_state()
        if glacier:
            sd0[C["SD_GLAC_CUM_MASS_BALANCE"], d.hru_iparams[C["HPI_IS_GLACIER"]] != 0] = 0.0
            m.set_state(sd0, si0)
    # the reference's own put_data next to its own steps; the device's put_data behind the binding
    outs = ["OUT_RUNOFF", "OUT_BASEFLOW", "OUT_EVAP", "OUT_SWE", "OUT_SOIL_MOIST", "OUT_PREC", "OUT_SWE_BAND"]
    a.put_data(-1); a.reset_agg()
    for s in range(nsteps):
        fr, cr, er = a.step(f[s], sf[s], dmy[s])
        assert er.sum() == 0
        a.put_data(s, f[s], cr, nsteps)
    want = np.concatenate([a.get_output(n, True) for n in outs]).astype(np.float32)
    flags, got = b.run_through_binding(f, sf, dmy, out_names=outs, out_step_ratio=nsteps, out_rows=want.shape[0])
    assert flags.sum() == 0
    from tests.util import rel_diff
    wo = rel_diff(want, got, 1e-4).max()
    assert wo < 1e-5, "aggregated outputs: the reference's put_data vs the device's behind the binding: %.3e" % wo
    (sa, ia), (sb, ib) = a.get_state(), b.get_state()
    sa[C["SD_ERROR"]] = 0; sb[C["SD_ERROR"]] = 0
    # free-running: near-zero node ice contents amplify the last bits of the node temperatures (tests/test_gpu_parity.py), so
    # all rows are compared with an absolute floor and the outputs north_star names relatively
    w, msg = worst(sa, sb, "SD_", floor=1e-2)
    head = [C[k] for k in ("SD_MOIST0", "SD_MOIST1", "SD_MOIST2", "SD_SNOW_SWQ", "SD_GLAC_CUM_MASS_BALANCE", "SD_GLAC_WATER_STORAGE")]
    w2, msg2 = worst(sa[head], sb[head], "SD_", floor=1e-4)
    print(name, "reference by itself vs through the binding on the GPU after %d steps: worst rel diff %.3e (headline rows %.3e)" % (nsteps, w, w2))
    assert w < 1e-5, msg
    assert w2 < 1e-6, msg2
    snow_rows = [C[r] for r in ("SI_SNOW_LAST_SNOW", "SI_SNOW_MELTING", "SI_SNOW_SNOW", "SI_SNOW_STORE_SNOW", "SI_FROZEN", "SI_NFROST", "SI_NTHAW")]
    assert np.array_equal(ia[snow_rows], ib[snow_rows])
    a.close(); b.close()
