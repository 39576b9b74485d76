"""Host-side binding of the C-ABI in include/vicgpu.h (libvicgpu.so, HIP/gfx950).

The method names mirror the reference driver's vocabulary for this path
(vicNl.c:390-654 runModel / dist_prec.c:8 dist_prec): a `Model` owns the domain
tables on one GPU, `push_forcing` replaces cell.atmos[rec], `dist_prec(rec0, n)`
advances every cell n records.  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, construction raises.
"""
import ctypes
import os

import numpy as np

from . import abi
from .abi import C

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvicgpu.so")

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_ubyte)

_lib = None


class VicGpuError(RuntimeError):
    pass


def load_library():
    """Loads libvicgpu.so and declares every entry point of include/vicgpu.h.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VICGPU_LIB", LIB_PATH)      # tuning: an alternative build of the same library (tools/ab.sh)
    if not os.path.exists(path):
        raise VicGpuError("HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    sig = {
        "vicgpu_abi_version": (ctypes.c_int, []),
        "vicgpu_create": (ctypes.c_int, [ctypes.POINTER(abi.Options), ctypes.c_int, ctypes.POINTER(vp)]),
        "vicgpu_destroy": (None, [vp]),
        "vicgpu_last_error": (ctypes.c_char_p, [vp]),
        "vicgpu_set_veglib": (ctypes.c_int, [vp, ctypes.c_int, _dp]),
        "vicgpu_set_domain": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, _dp, _ip, _dp, _ip, _ip]),
        "vicgpu_set_state": (ctypes.c_int, [vp, _dp, _ip]),
        "vicgpu_get_state": (ctypes.c_int, [vp, _dp, _ip]),
        "vicgpu_push_forcing": (ctypes.c_int, [vp, ctypes.c_int, _dp, _up, _ip]),
        "vicgpu_step": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int]),
        "vicgpu_synchronize": (ctypes.c_int, [vp]),
        "vicgpu_get_fluxes": (ctypes.c_int, [vp, _dp]),
        "vicgpu_get_cell_outputs": (ctypes.c_int, [vp, _dp]),
        "vicgpu_get_accum": (ctypes.c_int, [vp, _dp]),
        "vicgpu_reset_accum": (ctypes.c_int, [vp]),
        "vicgpu_get_cell_errors": (ctypes.c_int, [vp, _ip]),
        "vicgpu_set_stream": (ctypes.c_int, [vp, vp]),
        "vicgpu_set_write_fluxes": (ctypes.c_int, [vp, ctypes.c_int]),
        "vicgpu_device_ptr": (vp, [vp, ctypes.c_int]),
        "vicgpu_last_kernel_ms": (ctypes.c_int, [vp, _dp, _ip]),
        "vicgpu_debug_pure": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, _dp, _dp]),
        "vicgpu_glacier_mass_balance_fit": (ctypes.c_int, [vp, _dp, ctypes.c_int]),
        "vicgpu_prefetch_forcing": (ctypes.c_int, [vp, ctypes.c_int, _dp, _up, _ip]),
        "vicgpu_prefetch_forcing_raw": (ctypes.c_int, [vp, ctypes.c_int, _dp, _ip, ctypes.c_double, ctypes.c_int]),
        "vicgpu_swap_forcing": (ctypes.c_int, [vp]),
        "vicgpu_get_forcing": (ctypes.c_int, [vp, ctypes.c_int, _dp, _up]),
        "vicgpu_host_alloc": (vp, [ctypes.c_size_t]),
        "vicgpu_host_free": (None, [vp]),
        "vicgpu_get_state_records": (ctypes.c_int, [vp, _dp]),
        "vicgpu_set_state_records": (ctypes.c_int, [vp, _dp]),
        # include/vicgpu_out.h
        "vicgpu_out_nvar": (ctypes.c_int, []),
        "vicgpu_out_var_id": (ctypes.c_int, [ctypes.c_char_p]),
        "vicgpu_out_var_name": (ctypes.c_char_p, [ctypes.c_int]),
        "vicgpu_out_var_kind": (ctypes.c_int, [ctypes.c_int]),
        "vicgpu_out_var_agg": (ctypes.c_int, [ctypes.c_int]),
        "vicgpu_out_var_nelem": (ctypes.c_int, [ctypes.POINTER(abi.Options), ctypes.c_int]),
        "vicgpu_put_data_config": (ctypes.c_int, [vp, ctypes.c_int]),
        "vicgpu_put_data_init": (ctypes.c_int, [vp]),
        "vicgpu_get_outputs": (ctypes.c_int, [vp, ctypes.c_int, _ip, ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
        "vicgpu_get_output_data": (ctypes.c_int, [vp, ctypes.c_int, _ip, ctypes.c_int, _dp]),
        "vicgpu_get_balance": (ctypes.c_int, [vp, _dp]),
        "vicgpu_set_fluxes": (ctypes.c_int, [vp, _dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)   # AttributeError here = the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    if lib.vicgpu_abi_version() != C["VICGPU_ABI_VERSION"]:
        raise VicGpuError("ABI version mismatch between include/vicgpu.h and libvicgpu.so")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "vicgpu_abi_version", "vicgpu_create", "vicgpu_destroy", "vicgpu_last_error", "vicgpu_set_veglib", "vicgpu_set_domain",
    "vicgpu_set_state", "vicgpu_get_state", "vicgpu_push_forcing", "vicgpu_step", "vicgpu_synchronize", "vicgpu_get_fluxes",
    "vicgpu_get_cell_outputs", "vicgpu_get_accum", "vicgpu_reset_accum", "vicgpu_get_cell_errors", "vicgpu_set_stream",
    "vicgpu_set_write_fluxes", "vicgpu_device_ptr", "vicgpu_last_kernel_ms", "vicgpu_debug_pure",
    "vicgpu_glacier_mass_balance_fit",
    "vicgpu_out_nvar", "vicgpu_out_var_id", "vicgpu_out_var_name", "vicgpu_out_var_kind", "vicgpu_out_var_agg", "vicgpu_out_var_nelem",
    "vicgpu_put_data_config", "vicgpu_put_data_init", "vicgpu_get_outputs", "vicgpu_get_output_data", "vicgpu_get_balance",
    "vicgpu_set_fluxes", "vicgpu_get_state_records", "vicgpu_set_state_records",
    "vicgpu_prefetch_forcing", "vicgpu_prefetch_forcing_raw", "vicgpu_swap_forcing", "vicgpu_get_forcing", "vicgpu_host_alloc",
    "vicgpu_host_free",
]


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class Model:
    """One domain (or one rank's shard of it) resident on one GPU."""

    def __init__(self, dom, device=0):
        self.lib = load_library()
        self.dom = dom
        self.opt = dom.opt
        h = ctypes.c_void_p()
        rc = self.lib.vicgpu_create(ctypes.byref(dom.opt), device, ctypes.byref(h))
        if rc != 0:
            raise VicGpuError("vicgpu_create failed with code %d (no GPU, or unsupported options)" % rc)
        self.h = h
        self._chk(self.lib.vicgpu_set_veglib(h, dom.veglib.shape[0], _d(np.ascontiguousarray(dom.veglib))))
        self._chk(self.lib.vicgpu_set_domain(h, dom.ncell, dom.nhru, _d(dom.cell_params), _i(dom.hru_iparams),
                                             _d(dom.hru_dparams), _i(dom.cell_hru_offset), _i(dom.cell_hru_list)))

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.vicgpu_last_error(self.h)
            raise VicGpuError("vicgpu call failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vicgpu_destroy(self.h)
            self.h = None
            for p in getattr(self, "_pinned", []):
                self.lib.vicgpu_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state (initialize_model_state result in / write_model_state content out)
    def set_state(self, sd, si):
        sd = np.ascontiguousarray(sd, dtype=np.float64)
        si = np.ascontiguousarray(si, dtype=np.int32)
        assert sd.shape == (abi.sd_nrow(self.opt.Nnode), self.dom.nhru), sd.shape
        assert si.shape == (abi.si_nrow(self.opt.Nnode), self.dom.nhru), si.shape
        self._chk(self.lib.vicgpu_set_state(self.h, _d(sd), _i(si)))

    def get_state(self):
        sd = np.zeros((abi.sd_nrow(self.opt.Nnode), self.dom.nhru))
        si = np.zeros((abi.si_nrow(self.opt.Nnode), self.dom.nhru), dtype=np.int32)
        self._chk(self.lib.vicgpu_get_state(self.h, _d(sd), _i(si)))
        return sd, si

    # ---- forcing chunk
    def push_forcing(self, forcing, snowflag, dmy):
        forcing = np.ascontiguousarray(forcing, dtype=np.float64)
        snowflag = np.ascontiguousarray(snowflag, dtype=np.uint8)
        dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        n = forcing.shape[0]
        assert forcing.shape == (n, C["VIC_NFORCE"], self.opt.NR + 1, self.dom.ncell), forcing.shape
        assert snowflag.shape == (n, self.opt.NR + 1, self.dom.ncell)
        assert dmy.shape == (n, C["VIC_NDMY"])
        self._hold = (forcing, snowflag, dmy)   # the H2D copy is asynchronous
        self._chk(self.lib.vicgpu_push_forcing(self.h, n, _d(forcing), snowflag.ctypes.data_as(_up), _i(dmy)))

    # ---- forcing streaming: the next chunk uploads while the steps of the current one run
    def prefetch_forcing(self, forcing, snowflag, dmy):
        forcing = np.ascontiguousarray(forcing, dtype=np.float64)
        snowflag = np.ascontiguousarray(snowflag, dtype=np.uint8)
        dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        n = forcing.shape[0]
        assert forcing.shape == (n, C["VIC_NFORCE"], self.opt.NR + 1, self.dom.ncell), forcing.shape
        assert snowflag.shape == (n, self.opt.NR + 1, self.dom.ncell)
        self._hold_next = (forcing, snowflag, dmy)   # pinned sources are read until swap_forcing returns
        self._chk(self.lib.vicgpu_prefetch_forcing(self.h, n, _d(forcing), snowflag.ctypes.data_as(_up), _i(dmy)))

    def prefetch_forcing_raw(self, raw, dmy, min_wind_speed=0.0, plapse=True):
        """Hourly raw forcing [nsteps][VIC_NRAW][dt][ncell] (kPa pressures): initialize_atmos.c's derivation runs on the device."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        dmy = np.ascontiguousarray(dmy, dtype=np.int32)
        n = raw.shape[0]
        assert raw.shape == (n, C["VIC_NRAW"], self.opt.dt, self.dom.ncell), raw.shape
        self._hold_next = (raw, dmy)
        self._chk(self.lib.vicgpu_prefetch_forcing_raw(self.h, n, _d(raw), _i(dmy), float(min_wind_speed), int(bool(plapse))))

    def swap_forcing(self):
        self._chk(self.lib.vicgpu_swap_forcing(self.h))
        self._hold = getattr(self, "_hold_next", None)

    def get_forcing(self, step):
        f = np.zeros((C["VIC_NFORCE"], self.opt.NR + 1, self.dom.ncell)); sf = np.zeros((self.opt.NR + 1, self.dom.ncell), dtype=np.uint8)
        self._chk(self.lib.vicgpu_get_forcing(self.h, int(step), _d(f), sf.ctypes.data_as(_up)))
        return f, sf

    def pinned(self, shape, dtype=np.float64):
        """A numpy array over pinned host memory (vicgpu_host_alloc): forcing chunks in it are uploaded by DMA without a
        staging copy.  Freed when the Model is closed."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.vicgpu_host_alloc(n)
        if not p:
            raise VicGpuError("vicgpu_host_alloc(%d) failed" % n)
        self._pinned = getattr(self, "_pinned", []) + [p]
        buf = (ctypes.c_char * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    # ---- the hot path
    def dist_prec(self, rec0, nrec=1, sync=True):
        """dist_prec (dist_prec.c:8) for every cell, records [rec0, rec0+nrec) of the pushed chunk."""
        self._chk(self.lib.vicgpu_step(self.h, rec0, nrec))
        if sync:
            self._chk(self.lib.vicgpu_synchronize(self.h))

    def synchronize(self):
        self._chk(self.lib.vicgpu_synchronize(self.h))

    # ---- outputs
    def get_fluxes(self):
        fx = np.zeros((C["FX_NROW"], self.dom.nhru))
        self._chk(self.lib.vicgpu_get_fluxes(self.h, _d(fx)))
        return fx

    def get_cell_outputs(self):
        co = np.zeros((C["CO_NROW"], self.dom.ncell))
        self._chk(self.lib.vicgpu_get_cell_outputs(self.h, _d(co)))
        return co

    def get_accum(self):
        ac = np.zeros((C["CA_NROW"], self.dom.ncell))
        self._chk(self.lib.vicgpu_get_accum(self.h, _d(ac)))
        return ac

    def reset_accum(self):
        self._chk(self.lib.vicgpu_reset_accum(self.h))

    def get_cell_errors(self):
        ce = np.zeros(self.dom.ncell, dtype=np.int32)
        self._chk(self.lib.vicgpu_get_cell_errors(self.h, _i(ce)))
        return ce

    def set_write_fluxes(self, on):
        self._chk(self.lib.vicgpu_set_write_fluxes(self.h, int(bool(on))))

    # ---- the state as the reference's state file holds it (write_model_state.c:95-337)
    def get_state_records(self):
        rec = np.zeros((self.dom.nhru, abi.sr_len(self.opt.Nnode)))
        self._chk(self.lib.vicgpu_get_state_records(self.h, _d(rec)))
        return rec

    def set_state_records(self, rec):
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        assert rec.shape == (self.dom.nhru, abi.sr_len(self.opt.Nnode))
        self._chk(self.lib.vicgpu_set_state_records(self.h, _d(rec)))

    # ---- put_data: the aggregated output variables (include/vicgpu_out.h)
    def output_list(self):
        """[(name, nelem, aggregation)] of every variable the library provides, index = variable id."""
        lib = self.lib
        return [(lib.vicgpu_out_var_name(v).decode(), lib.vicgpu_out_var_nelem(ctypes.byref(self.opt), v), lib.vicgpu_out_var_agg(v))
                for v in range(lib.vicgpu_out_nvar())]

    def var_ids(self, names):
        ids = [self.lib.vicgpu_out_var_id(n.encode()) for n in names]
        if min(ids) < 0:
            raise VicGpuError("output variable not provided: %s" % [n for n, i in zip(names, ids) if i < 0])
        return np.asarray(ids, dtype=np.int32)

    def put_data_config(self, out_step_ratio=1):
        self._chk(self.lib.vicgpu_put_data_config(self.h, int(out_step_ratio)))

    def put_data_init(self):
        self._chk(self.lib.vicgpu_put_data_init(self.h))

    def _out_rows(self, ids):
        return int(sum(self.lib.vicgpu_out_var_nelem(ctypes.byref(self.opt), int(v)) for v in ids))

    def get_outputs(self, names, reset=True):
        """The aggregates (OutputData.aggdata) of the named variables as the writer wants them: float32 [sum nelem][ncell]."""
        ids = self.var_ids(names)
        out = np.zeros((self._out_rows(ids), self.dom.ncell), dtype=np.float32)
        self._chk(self.lib.vicgpu_get_outputs(self.h, len(ids), _i(ids), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(bool(reset))))
        return out

    def get_output_data(self, names, aggregated=False):
        ids = self.var_ids(names)
        out = np.zeros((self._out_rows(ids), self.dom.ncell))
        self._chk(self.lib.vicgpu_get_output_data(self.h, len(ids), _i(ids), int(bool(aggregated)), _d(out)))
        return out

    def get_balance(self):
        pb = np.zeros((C["PB_NROW"], self.dom.ncell))
        self._chk(self.lib.vicgpu_get_balance(self.h, _d(pb)))
        return pb

    def set_fluxes(self, fx):
        fx = np.ascontiguousarray(fx, dtype=np.float64)
        assert fx.shape == (C["FX_NROW"], self.dom.nhru)
        self._chk(self.lib.vicgpu_set_fluxes(self.h, _d(fx)))

    def glacier_mass_balance_fit(self, reset=True):
        """End of a glacier accumulation interval (accumulateGlacierMassBalance.c:53-66): [GMB_NROW][ncell] fit per cell."""
        eq = np.zeros((C["GMB_NROW"], self.dom.ncell))
        self._chk(self.lib.vicgpu_glacier_mass_balance_fit(self.h, _d(eq), int(bool(reset))))
        return eq

    def debug_pure(self, fn, inputs):
        """Test hook (vicgpu_debug_pure): one pure function of the path for every row of inputs [n][VICGPU_PURE_NIN]."""
        inp = np.zeros((len(inputs), C["VICGPU_PURE_NIN"]) if "VICGPU_PURE_NIN" in C else (len(inputs), 10))
        inputs = np.asarray(inputs, dtype=np.float64)
        inp[:, :inputs.shape[1]] = inputs
        out = np.zeros(inp.shape[0])
        self._chk(self.lib.vicgpu_debug_pure(self.h, int(fn), inp.shape[0], _d(inp), _d(out)))
        return out

    def last_kernel_ms(self):
        ms = ctypes.c_double(0)
        n = ctypes.c_int(0)
        self._chk(self.lib.vicgpu_last_kernel_ms(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
