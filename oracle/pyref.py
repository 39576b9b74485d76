"""ctypes wrappers around the checker libraries — TEST INFRASTRUCTURE ONLY.

RefModel   -> oracle/_ref/libvicref*.so   (the real reference, built by oracle/ref_build/build_ref.sh)
OracleModel-> oracle/libvicoracle.so       (the C restatement, oracle/orc_*.c)

Both expose the same small interface over the tables of include/vicgpu.h so that
tests can drive reference, oracle and the HIP product with identical inputs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os

import numpy as np

from vic_amd import abi
from vic_amd.abi import C

_HERE = os.path.dirname(os.path.abspath(__file__))

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_ubyte)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _u(a):
    return None if a is None else a.ctypes.data_as(_up)


def ref_lib_path(variant="plain"):
    name = "libvicref.so" if variant == "plain" else "libvicref_%s.so" % variant
    return os.path.join(_HERE, "_ref", name)


def oracle_lib_path():
    return os.path.join(_HERE, "libvicoracle.so")


class _Model:
    prefix = None

    def __init__(self, lib, dom):
        self.lib = lib
        self.dom = dom
        self.opt = dom.opt
        p = self.prefix
        f = getattr(lib, p + "create"); f.restype = ctypes.c_void_p; f.argtypes = [ctypes.POINTER(abi.Options)]
        self.h = f(ctypes.byref(dom.opt))
        if not self.h:
            raise RuntimeError("%screate failed" % p)
        self._fn("set_veglib", [ctypes.c_void_p, ctypes.c_int, _dp])
        self._fn("set_domain", [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _ip, _dp, _ip, _ip])
        self._fn("get_state", [ctypes.c_void_p, _dp, _ip])
        self._fn("set_state", [ctypes.c_void_p, _dp, _ip])
        self._fn("step", [ctypes.c_void_p, _dp, _up, _ip, _dp, _dp, _ip, ctypes.c_int])
        r = getattr(lib, p + "run"); r.restype = ctypes.c_double
        r.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, _up, _ip, ctypes.c_int]
        rc = getattr(lib, p + "set_veglib")(self.h, dom.veglib.shape[0], _d(dom.veglib))
        assert rc == 0, rc
        rc = getattr(lib, p + "set_domain")(self.h, dom.ncell, dom.nhru, _d(dom.cell_params), _i(dom.hru_iparams),
                                            _d(dom.hru_dparams), _i(dom.cell_hru_offset), _i(dom.cell_hru_list))
        assert rc == 0, rc

    def _fn(self, name, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = ctypes.c_int
        f.argtypes = argtypes

    def close(self):
        if self.h:
            d = getattr(self.lib, self.prefix + "destroy"); d.restype = None; d.argtypes = [ctypes.c_void_p]
            d(self.h)
            self.h = None

    def get_state(self):
        sd = np.zeros((abi.sd_nrow(self.opt.Nnode), self.dom.nhru))
        si = np.zeros((abi.si_nrow(self.opt.Nnode), self.dom.nhru), dtype=np.int32)
        rc = getattr(self.lib, self.prefix + "get_state")(self.h, _d(sd), _i(si))
        assert rc == 0
        return sd, si

    def set_state(self, sd, si):
        sd = np.ascontiguousarray(sd, dtype=np.float64); si = np.ascontiguousarray(si, dtype=np.int32)
        rc = getattr(self.lib, self.prefix + "set_state")(self.h, _d(sd), _i(si))
        assert rc == 0

    def step(self, forcing, snowflag, dmy, nthreads=1):
        """One step: forcing [NFORCE][NF+1][ncell]. Returns (flux, cell_out, cell_err)."""
        forcing = np.ascontiguousarray(forcing); snowflag = np.ascontiguousarray(snowflag)
        dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        fx = np.zeros((C["FX_NROW"], self.dom.nhru))
        co = np.zeros((C["CO_NROW"], self.dom.ncell))
        ce = np.zeros(self.dom.ncell, dtype=np.int32)
        getattr(self.lib, self.prefix + "step")(self.h, _d(forcing), _u(snowflag), _i(dmy), _d(fx), _d(co), _i(ce), nthreads)
        return fx, co, ce

    def glacier_fit(self, reset=True):
        eq = np.zeros((C["GMB_NROW"], self.dom.ncell))
        f = getattr(self.lib, self.prefix + "glacier_fit")
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int]
        rc = f(self.h, _d(eq), int(bool(reset)))
        assert rc == 0, rc
        return eq

    def pure(self, fn, inputs):
        """Test hook (<prefix>pure): one pure function of the path for every row of inputs [n][<= 10]."""
        inputs = np.asarray(inputs, dtype=np.float64)
        inp = np.zeros((inputs.shape[0], 10))
        inp[:, :inputs.shape[1]] = inputs
        out = np.zeros(inp.shape[0])
        f = getattr(self.lib, self.prefix + "pure")
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp]
        rc = f(self.h, int(fn), inp.shape[0], _d(inp), _d(out))
        assert rc == 0, rc
        return out

    # ---- put_data (oracle/orc_putdata.c, the reference's put_data.c through the shim)
    def get_fluxes(self):
        fx = np.zeros((C["FX_NROW"], self.dom.nhru))
        f = getattr(self.lib, self.prefix + "get_fluxes"); f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp]
        assert f(self.h, _d(fx)) == 0
        return fx

    def put_data(self, rec, forcing=None, cell_out=None, out_step_ratio=1):
        """put_data for every cell after a step (rec >= 0) or the initialisation call before the first one (rec < 0)."""
        f = getattr(self.lib, self.prefix + "put_data"); f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, _dp, ctypes.c_int]
        forcing = None if forcing is None else np.ascontiguousarray(forcing)
        cell_out = None if cell_out is None else np.ascontiguousarray(cell_out)
        rc = f(self.h, int(rec), _d(forcing), _d(cell_out), int(out_step_ratio))
        assert rc == 0, rc

    def get_balance(self):
        pb = np.zeros((C["PB_NROW"], self.dom.ncell))
        f = getattr(self.lib, self.prefix + "get_balance"); f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp]
        assert f(self.h, _d(pb)) == 0
        return pb

    def reset_agg(self):
        f = getattr(self.lib, self.prefix + "reset_agg"); f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p]
        assert f(self.h) == 0

    def run(self, forcing, snowflag, dmy, nthreads=1):
        forcing = np.ascontiguousarray(forcing); snowflag = np.ascontiguousarray(snowflag)
        dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        return getattr(self.lib, self.prefix + "run")(self.h, forcing.shape[0], _d(forcing), _u(snowflag), _i(dmy), nthreads)


class RefModel(_Model):
    prefix = "vicref_"

    def __init__(self, dom, variant="plain"):
        path = ref_lib_path(variant)
        lib = ctypes.CDLL(path, mode=os.RTLD_LAZY)   # lazily bound: see build_ref.sh
        super().__init__(lib, dom)
        self._fn("init_state", [ctypes.c_void_p, _dp, _ip, _dp])
        self._fn("get_cell_params", [ctypes.c_void_p, _dp])

    def init_state(self, forcing0, dmy0, init_moist):
        forcing0 = np.ascontiguousarray(forcing0); dmy0 = np.ascontiguousarray(dmy0, dtype=np.int32)
        init_moist = np.ascontiguousarray(init_moist)
        rc = self.lib.vicref_init_state(self.h, _d(forcing0), _i(dmy0), _d(init_moist))
        assert rc == 0, rc

    def get_cell_params(self):
        out = np.zeros_like(self.dom.cell_params)
        self.lib.vicref_get_cell_params(self.h, _d(out))
        return out

    def derive_forcing(self, file, force_dt, min_wind=0.0, plapse=1):
        """The reference's own initialize_atmos (initialize_atmos.c:7-1349) on forcing records held in memory
        (oracle/ref_build/vicref_shim.cpp: the harness is the file reader, everything else is the reference).
        file [VIC_NRAW][nsteps * dt / force_dt][ncell] in file units (kPa); returns (forcing, snowflag) as vicgpu_push_forcing takes them."""
        file = np.ascontiguousarray(file, dtype=np.float64)
        o = self.dom.opt
        nsteps = file.shape[1] * force_dt // o.dt
        nsub = o.NR + 1
        f = np.zeros((nsteps, C["VIC_NFORCE"], nsub, self.dom.ncell)); sf = np.zeros((nsteps, nsub, self.dom.ncell), dtype=np.uint8)
        fn = self.lib.vicref_derive_forcing; fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_double, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte)]
        rc = fn(self.h, nsteps, int(force_dt), _d(file), float(min_wind), int(plapse), _d(f), sf.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        assert rc == 0, rc
        return f, sf

    def output_list(self):
        """The reference's own output variable list: {name: (index, nelem, aggtype)} (create_output_list)."""
        self.lib.vicref_out_nvar.restype = ctypes.c_int
        f = self.lib.vicref_out_info; f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, _ip, _ip]
        out = {}
        for v in range(self.lib.vicref_out_nvar()):
            name = ctypes.create_string_buffer(64); ne = ctypes.c_int(0); ag = ctypes.c_int(0)
            assert f(self.h, v, name, ctypes.byref(ne), ctypes.byref(ag)) == 0
            out[name.value.decode()] = (v, ne.value, ag.value)
        return out

    def get_cell_params(self):
        """The cell table as the reference holds it after initialize_model_state: the node geometry / node constants are
        recomputed there (set_node_parameters, soil_conduction.c:142-303) and may differ from the generator's in the last bit."""
        from vic_amd import abi
        out = np.zeros((abi.cp_nrow(self.dom.opt.Nnode, self.dom.opt.Nband), self.dom.ncell))
        f = self.lib.vicref_get_cell_params; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp]
        assert f(self.h, _d(out)) == 0
        return out

    def binding_tables(self):
        """What integration/vicgpu_binding.cpp packs from the harness's reference structs: dict of its tables (its own HRU
        numbering) + the vicgpu_options it derives from ProgramState."""
        from vic_amd import abi
        d, o = self.dom, self.dom.opt
        t = dict(veglib=np.zeros((o.nveg_types + 4, C["VL_NFIELD"])), cell_params=np.zeros((abi.cp_nrow(o.Nnode, o.Nband), d.ncell)),
                 hpi=np.zeros((C["HPI_NROW"], d.nhru), dtype=np.int32), hpd=np.zeros((C["HPD_NROW"], d.nhru)),
                 cell_off=np.zeros(d.ncell + 1, dtype=np.int32), cell_list=np.zeros(d.nhru, dtype=np.int32),
                 sd=np.zeros((abi.sd_nrow(o.Nnode), d.nhru)), si=np.zeros((abi.si_nrow(o.Nnode), d.nhru), dtype=np.int32), opt=abi.Options())
        f = self.lib.vicref_binding_tables; f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, _dp, _dp, _ip, _dp, _ip, _ip, _dp, _ip, ctypes.POINTER(abi.Options)]
        assert f(self.h, _d(t["veglib"]), _d(t["cell_params"]), _i(t["hpi"]), _d(t["hpd"]), _i(t["cell_off"]), _i(t["cell_list"]),
                 _d(t["sd"]), _i(t["si"]), ctypes.byref(t["opt"])) == 0
        return t

    def run_through_binding(self, forcing, snowflag, dmy, device=0, out_names=(), out_step_ratio=1, out_rows=0):
        """All steps through VicGpuBinding (the reference-side binding) on the GPU; libvicgpu.so is loaded globally first so
        that the binding's vicgpu_* calls resolve.  out_names: put_data runs on the device as well and the aggregates of
        those variables come back (float32 [out_rows][ncell]).  Returns (per-cell error flags, outputs or None)."""
        from vic_amd import api
        ctypes.CDLL(os.environ.get("VICGPU_LIB", api.LIB_PATH), mode=os.RTLD_GLOBAL | os.RTLD_NOW)
        forcing = np.ascontiguousarray(forcing); snowflag = np.ascontiguousarray(snowflag); dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        flags = np.zeros(self.dom.ncell, dtype=np.int32)
        outs = np.zeros((max(out_rows, 1), self.dom.ncell), dtype=np.float32)
        names = (ctypes.c_char_p * max(len(out_names), 1))(*[n.encode() for n in out_names])
        f = self.lib.vicref_run_through_binding; f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte), _ip, ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int,
                      ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float)]
        rc = f(self.h, forcing.shape[0], _d(forcing), snowflag.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _i(dmy), int(device), _i(flags),
               int(out_step_ratio), len(out_names), names, outs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        assert rc == 0, "binding returned %d" % rc
        return flags, (outs if out_names else None)

    def state_stream(self):
        """The reference's own state-file stream of every cell (processCellForStateFile into a memory back-end):
        (values, variable ids, cell_start[ncell + 1])."""
        f = self.lib.vicref_state_stream_write; f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, _dp, _ip, ctypes.c_int, _ip]
        n = f(self.h, None, None, 0, None)
        vals = np.zeros(n); ids = np.zeros(n, dtype=np.int32); start = np.zeros(self.dom.ncell + 1, dtype=np.int32)
        assert f(self.h, _d(vals), _i(ids), n, _i(start)) == n
        return vals, ids, start

    def read_state_stream(self, vals, ids):
        f = self.lib.vicref_state_stream_read; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp, _ip, ctypes.c_int]
        vals = np.ascontiguousarray(vals, dtype=np.float64); ids = np.ascontiguousarray(ids, dtype=np.int32)
        return f(self.h, _d(vals), _i(ids), len(vals))

    def state_var_id(self, name):
        f = self.lib.vicref_state_var_id; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_char_p]
        return f(name.encode())

    def get_output(self, name, agg=False):
        """OutputData.data (or .aggdata) of a variable of the reference's list by name: [nelem][ncell]."""
        v, ne, _ = self.output_list()[name]
        out = np.zeros((ne, self.dom.ncell))
        f = self.lib.vicref_get_output; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
        assert f(self.h, v, int(agg), _d(out)) == ne
        return out


class OracleModel(_Model):
    prefix = "vicorc_"

    def derive_forcing(self, raw, min_wind=0.0, plapse=1):
        """initialize_atmos.c's derivation of atmos[rec] from hourly raw forcing [nsteps][VIC_NRAW][dt][ncell]."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        nsub = self.dom.opt.NR + 1
        f = np.zeros((raw.shape[0], C["VIC_NFORCE"], nsub, self.dom.ncell)); sf = np.zeros((raw.shape[0], nsub, self.dom.ncell), dtype=np.uint8)
        fn = self.lib.vicorc_derive_forcing; fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.c_double, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte)]
        assert fn(self.h, raw.shape[0], _d(raw), float(min_wind), int(plapse), _d(f), sf.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))) == 0
        return f, sf

    def implicit_stats(self):
        """(converged, failed) calls of the implicit soil heat solver so far."""
        ok, bad = ctypes.c_long(0), ctypes.c_long(0)
        f = self.lib.vicorc_implicit_stats; f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        f(self.h, ctypes.byref(ok), ctypes.byref(bad))
        return ok.value, bad.value

    def get_state_records(self):
        from vic_amd import abi
        rec = np.zeros((self.dom.nhru, abi.sr_len(self.dom.opt.Nnode)))
        f = self.lib.vicorc_state_records; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int]
        assert f(self.h, _d(rec), 1) == 0
        return rec

    def set_state_records(self, rec):
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        f = self.lib.vicorc_state_records; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int]
        return f(self.h, _d(rec), 0)

    def __init__(self, dom, converged_nodes=False):
        """converged_nodes: the frozen-node root finds (soil_thermal_eqn.c) iterate to 1e-13 K instead of the reference's
        1e-7 K -- the checker for the product's Newton node solver, which converges those roots fully (orc.h)."""
        lib = ctypes.CDLL(oracle_lib_path())
        super().__init__(lib, dom)
        if converged_nodes:
            lib.vicorc_set_node_tolerance.restype = ctypes.c_int
            lib.vicorc_set_node_tolerance.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
            assert lib.vicorc_set_node_tolerance(self.h, 1e-16, 1e-13) == 0


def _orc_out(self):
    self.lib.vicorc_out_var_id.restype = ctypes.c_int; self.lib.vicorc_out_var_id.argtypes = [ctypes.c_char_p]
    self.lib.vicorc_out_var_nelem.restype = ctypes.c_int; self.lib.vicorc_out_var_nelem.argtypes = [ctypes.c_void_p, ctypes.c_int]
    self.lib.vicorc_out_var_agg.restype = ctypes.c_int; self.lib.vicorc_out_var_agg.argtypes = [ctypes.c_int]
    self.lib.vicorc_out_var_name.restype = ctypes.c_char_p; self.lib.vicorc_out_var_name.argtypes = [ctypes.c_int]
    self.lib.vicorc_out_nvar.restype = ctypes.c_int


def _orc_output_list(self):
    """{name: (index, nelem, aggregation)} of the variables the oracle (and the product) provides."""
    _orc_out(self)
    return {self.lib.vicorc_out_var_name(v).decode(): (v, self.lib.vicorc_out_var_nelem(self.h, v), self.lib.vicorc_out_var_agg(v))
            for v in range(self.lib.vicorc_out_nvar())}


def _orc_get_output(self, name, agg=False):
    v, ne, _ = self.output_list()[name]
    out = np.zeros((ne, self.dom.ncell))
    f = self.lib.vicorc_get_output; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    assert f(self.h, v, int(agg), _d(out)) == ne
    return out


def _orc_set_fluxes(self, fx):
    fx = np.ascontiguousarray(fx, dtype=np.float64)
    f = self.lib.vicorc_set_fluxes; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, _dp]
    assert f(self.h, _d(fx)) == 0


OracleModel.output_list = _orc_output_list
OracleModel.get_output = _orc_get_output
OracleModel.set_fluxes = _orc_set_fluxes


def have_ref(variant="plain"):
    return os.path.exists(ref_lib_path(variant))


def have_oracle():
    return os.path.exists(oracle_lib_path())
