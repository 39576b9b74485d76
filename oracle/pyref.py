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


class OracleModel(_Model):
    prefix = "vicorc_"

    def __init__(self, dom, converged_nodes=False):
        """converged_nodes: the frozen-node root finds (soil_thermal_eqn.c) iterate to 1e-13 K instead of the reference's
        1e-7 K -- the checker for the product's Newton node solver, which converges those roots fully (orc.h)."""
        lib = ctypes.CDLL(oracle_lib_path())
        super().__init__(lib, dom)
        if converged_nodes:
            lib.vicorc_set_node_tolerance.restype = ctypes.c_int
            lib.vicorc_set_node_tolerance.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
            assert lib.vicorc_set_node_tolerance(self.h, 1e-16, 1e-13) == 0


def have_ref(variant="plain"):
    return os.path.exists(ref_lib_path(variant))


def have_oracle():
    return os.path.exists(oracle_lib_path())
