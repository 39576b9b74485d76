"""Host-side ingest logic (vic_amd/domain.py, vic_amd/init_state.py) against the reference's initialize_model_state."""
import numpy as np
import pytest

from vic_amd import abi, domain, init_state, shard
from vic_amd.abi import C
from tests.golden_util import golden_names, load_golden
from tests.util import rel_diff


@pytest.mark.parametrize("kw,variant", [
    (dict(FULL_ENERGY=1), "plain"),
    (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10), "fixed"),
    (dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=14, Nband=3), "fixed"),
    (dict(FULL_ENERGY=0, dt=24, snow_step=3), "plain"),
])
def test_initial_state_matches_reference(kw, variant, oracle_lib, ref_available):
    if not ref_available:
        pytest.skip("reference build (oracle/_ref) not available")
    opt = abi.default_options(**kw)
    d = domain.make_domain(40, opt, ntile=3)
    f, sf, dmy = domain.make_forcing(d, 0, 1, start_doy=1)
    ref = oracle_lib.RefModel(d, variant)
    ref.init_state(f[0], dmy[0], d.init_moist)
    # node geometry / node constants written by initialize_model_state -> set_node_parameters
    assert np.array_equal(ref.get_cell_params(), d.cell_params, equal_nan=True)
    sr, ir = ref.get_state()
    sp, ip = init_state.initial_state(d, f[0])
    assert rel_diff(sr, sp, 1e-12).max() < 1e-12
    assert np.array_equal(ir, ip)
    ref.close()


@pytest.mark.parametrize("name", golden_names())
def test_initial_state_matches_golden(name):
    """Same check against the committed fixtures (runs without the reference)."""
    d, z = load_golden(name)
    sp, ip = init_state.initial_state(d, z["forcing"][0])
    sd0 = z["sd0"].copy()
    if (d.hru_iparams[C["HPI_IS_GLACIER"]] != 0).any():
        isg = d.hru_iparams[C["HPI_IS_GLACIER"]] != 0
        sp[C["SD_GLAC_CUM_MASS_BALANCE"], isg] = 0.0   # the fixture opens the accumulation window
    assert rel_diff(sd0, sp, 1e-12).max() < 1e-12
    assert np.array_equal(z["si0"], ip)


def test_forcing_conventions():
    opt = abi.default_options(FULL_ENERGY=0, dt=24, snow_step=3)
    d = domain.make_domain(16, opt, ntile=1)
    f, sf, dmy = domain.make_forcing(d, 0, 5, start_doy=360)
    NF, NR = opt.NF, opt.NR
    assert (NF, NR) == (8, 8)
    assert f.shape == (5, C["VIC_NFORCE"], 9, 16)
    # step value = mean of the sub-steps (sum for precipitation); snowflag[NR] = any sub-step flag
    assert np.allclose(f[:, C["VIC_F_AIR_TEMP"], NR], f[:, C["VIC_F_AIR_TEMP"], :NF].mean(axis=1))
    assert np.allclose(f[:, C["VIC_F_PREC"], NR], f[:, C["VIC_F_PREC"], :NF].sum(axis=1))
    assert np.array_equal(sf[:, NR], sf[:, :NF].max(axis=1))
    assert np.allclose(f[:, C["VIC_F_VPD"], :NF], domain.svp(f[:, C["VIC_F_AIR_TEMP"], :NF]) - f[:, C["VIC_F_VP"], :NF])
    assert dmy[0, C["VIC_DMY_MONTH"]] == 12 and dmy[-1, C["VIC_DMY_DAY_IN_YEAR"]] == 364


def test_shard_domain_partitions_cells():
    opt = abi.default_options(FULL_ENERGY=1, Nband=2)
    d = domain.make_domain(37, opt, ntile=3)
    b = shard.partition_cells(d.cell_hru_offset, 4)
    assert b[0] == 0 and b[-1] == 37 and (np.diff(b) > 0).all()
    seen = []
    for r in range(4):
        s = shard.shard_domain(d, r, 4)
        assert s.cell_hru_offset[0] == 0 and s.cell_hru_offset[-1] == s.nhru
        assert (s.hru_iparams[C["HPI_CELL"]] >= 0).all() and (s.hru_iparams[C["HPI_CELL"]] < s.ncell).all()
        # every shard HRU keeps its parameters
        g = s.global_hru_ids
        assert np.array_equal(s.hru_dparams, d.hru_dparams[:, g])
        assert np.array_equal(s.cell_params, d.cell_params[:, s.global_cell0:s.global_cell0 + s.ncell], equal_nan=True)
        seen.append(g)
    allg = np.sort(np.concatenate(seen))
    assert np.array_equal(allg, np.arange(d.nhru))


def test_cell_range_equals_shard_of_the_full_domain():
    """bench.py's multi-GPU layout: every rank builds only its block of the one big domain
    (domain.make_domain(cell_range=...)); that must be exactly what shard.shard_domain cuts out of the full domain --
    parameter tables, HRU lists, initial moisture and the forcing of the block's cells."""
    from vic_amd import shard
    kw = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=5)
    d = domain.make_domain(97, abi.default_options(**kw), ntile=5, glacier_top_band=True)
    world = 4
    b = shard.partition_cells(d.cell_hru_offset, world)
    fg = domain.make_forcing(d, 0, 3, start_doy=60)
    for r in range(world):
        s1 = shard.shard_domain(d, r, world)
        s2 = domain.make_domain(97, abi.default_options(**kw), ntile=5, glacier_top_band=True, cell_range=(b[r], b[r + 1]))
        for k in ("cell_params", "hru_iparams", "hru_dparams", "cell_hru_offset", "cell_hru_list", "init_moist"):
            assert np.array_equal(getattr(s1, k), getattr(s2, k), equal_nan=True), (r, k)
        f1 = domain.make_forcing(s1, 0, 3, start_doy=60)
        f2 = domain.make_forcing(s2, 0, 3, start_doy=60)
        assert all(np.array_equal(x, y) for x, y in zip(f1, f2))
        assert np.array_equal(f1[0], fg[0][..., b[r]:b[r + 1]]) and np.array_equal(f1[1], fg[1][..., b[r]:b[r + 1]])
