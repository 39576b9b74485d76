"""Dev tool: free-running comparison reference vs oracle on a small domain."""
import sys, numpy as np
sys.path.insert(0, '.')
from vic_amd import abi, domain
from vic_amd.abi import C
from oracle.pyref import RefModel, OracleModel

def names(prefix):
    return {v: k for k, v in C.items() if k.startswith(prefix)}

def run(opt_kw, ncell=6, ntile=3, nsteps=240, start_doy=1, variant="plain", cold=0.0, glacier=False, verbose=True):
    opt = abi.default_options(**opt_kw)
    d = domain.make_domain(ncell, opt, ntile=ntile, glacier_top_band=glacier)
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=start_doy, cold=cold)
    ref = RefModel(d, variant)
    ref.init_state(f[0], dmy[0], d.init_moist)
    d.cell_params = ref.get_cell_params()
    sd0, si0 = ref.get_state()
    orc = OracleModel(d)
    orc.set_state(sd0, si0)
    sdn = names("SD_"); fxn = names("FX_")
    worst = 0.0
    for s in range(nsteps):
        fr, cr, er = ref.step(f[s], sf[s], dmy[s])
        fo, co, eo = orc.step(f[s], sf[s], dmy[s])
        sr, ir = ref.get_state(); so, io = orc.get_state()
        def rel(a, b):
            with np.errstate(all='ignore'):
                dd = np.abs(a - b) / np.maximum(1e-9, np.maximum(np.abs(a), np.abs(b)))
            dd = np.where(np.isnan(a) & np.isnan(b), 0, dd)
            dd = np.where(np.isnan(dd), np.inf, dd)
            dd = np.where((a == b), 0, dd)
            return dd
        ds = rel(sr, so); rows = [r for r in range(C["FX_NROW"]) if r not in (C["FX_OUT_PREC"], C["FX_OUT_RAIN"], C["FX_OUT_SNOW"])]; df = rel(fr[rows], fo[rows]); dc = rel(cr, co)
        m = max(ds.max(), df.max(), dc.max(), float((ir != io).any()))
        worst = max(worst, m)
        if m > 1e-9 and verbose:
            print("step", s, "max rel diff state %.3e flux %.3e cell %.3e int %d" % (ds.max(), df.max(), dc.max(), (ir != io).sum()))
            r, c = np.unravel_index(np.argmax(ds), ds.shape)
            print("   state row", r, sdn.get(r, "node+%d" % (r - C["SD_NSCALAR"])), "hru", c, sr[r, c], so[r, c])
            r, c = np.unravel_index(np.argmax(df), df.shape)
            print("   flux row", rows[r], fxn.get(rows[r]), "hru", c, fr[rows[r], c], fo[rows[r], c])
            if (ir != io).any():
                r, c = np.argwhere(ir != io)[0]
                print("   int row", r, "hru", c, ir[r, c], io[r, c])
            break
    print("steps", s + 1, "worst rel diff %.3e" % worst, "swq max", sr[C["SD_SNOW_SWQ"]].max(), "errs", er.sum(), eo.sum())
    return worst

if __name__ == "__main__":
    run(dict(FULL_ENERGY=1), nsteps=int(sys.argv[1]) if len(sys.argv) > 1 else 240)
