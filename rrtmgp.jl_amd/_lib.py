"""Loader for libhip_rrtmgp.so (the HIP back end).  There is no CPU fallback: if the
library is missing, cannot be loaded, or sees no GPU, the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RRTMGP_HIP_LIBRARY selects another build of the same library (the raw-instruction Float32 build
# `libhip_rrtmgp_fast.so`, or an A/B variant under variants/); the default is the shipped one.
SO_PATH = os.environ.get("RRTMGP_HIP_LIBRARY") or os.path.join(_HERE, "libhip_rrtmgp.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None

_P = C.c_void_p
_V2 = C.POINTER(_abi.View2D)
_SOLVE_SPECTRAL = [_P, _P, _P, _P, C.POINTER(_abi.AtmosState), _P, C.POINTER(_abi.FluxOut), C.POINTER(_abi.SolveOpts)]

EXPORTS = {
    # name: (restype, argtypes)
    "rrtmgp_hip_device_count": (C.c_int, []),
    "rrtmgp_hip_gas_lookup_create": (C.c_int, [C.POINTER(_abi.GasLookupDesc), C.c_int, C.POINTER(_P)]),
    "rrtmgp_hip_cloud_lookup_create": (C.c_int, [C.POINTER(_abi.CloudLookupDesc), C.c_int, C.POINTER(_P)]),
    "rrtmgp_hip_aerosol_lookup_create": (C.c_int, [C.POINTER(_abi.AerosolLookupDesc), C.c_int, C.POINTER(_P)]),
    "rrtmgp_hip_lookup_destroy": (C.c_int, [_P]),
    "rrtmgp_hip_workspace_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "rrtmgp_hip_workspace_destroy": (C.c_int, [_P]),
    "rrtmgp_hip_workspace_set_stream": (C.c_int, [_P, _P]),
    "rrtmgp_hip_workspace_synchronize": (C.c_int, [_P]),
    "rrtmgp_hip_workspace_last_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "rrtmgp_hip_rte_lw_2stream_solve": (C.c_int, _SOLVE_SPECTRAL),
    "rrtmgp_hip_rte_lw_noscat_solve": (C.c_int, _SOLVE_SPECTRAL),
    "rrtmgp_hip_rte_sw_2stream_solve": (C.c_int, _SOLVE_SPECTRAL),
    "rrtmgp_hip_rte_sw_noscat_solve": (C.c_int, [_P, _P, C.POINTER(_abi.AtmosState), _P, C.POINTER(_abi.FluxOut),
                                                 C.POINTER(_abi.SolveOpts)]),
    "rrtmgp_hip_rte_lw_2stream_solve_gray": (C.c_int, [_P, C.POINTER(_abi.GrayState), _P, C.POINTER(_abi.FluxOut),
                                                       C.POINTER(_abi.SolveOpts)]),
    "rrtmgp_hip_rte_lw_noscat_solve_gray": (C.c_int, [_P, C.POINTER(_abi.GrayState), _P, C.POINTER(_abi.FluxOut),
                                                      C.POINTER(_abi.SolveOpts)]),
    "rrtmgp_hip_rte_sw_2stream_solve_gray": (C.c_int, [_P, C.POINTER(_abi.GrayState), _P, C.POINTER(_abi.FluxOut),
                                                       C.POINTER(_abi.SolveOpts)]),
    "rrtmgp_hip_rte_sw_noscat_solve_gray": (C.c_int, [_P, C.POINTER(_abi.GrayState), _P, C.POINTER(_abi.FluxOut),
                                                      C.POINTER(_abi.SolveOpts)]),
    "rrtmgp_hip_compute_col_gas": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _V2, _V2, C.POINTER(_abi.Params), _V2, _P]),
    "rrtmgp_hip_compute_relative_humidity": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _V2, _V2, _V2,
                                                       C.POINTER(_abi.Params), _V2]),
    "rrtmgp_hip_compute_gray_heating_rate": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _V2, _V2, _V2, C.c_double,
                                                       C.c_double]),
    "rrtmgp_hip_prepare_atmosphere": (C.c_int, [_P, C.POINTER(_abi.AtmosState), C.POINTER(_abi.Params),
                                                C.POINTER(_abi.PrepareOpts)]),
    "rrtmgp_hip_prepare_atmosphere_gray": (C.c_int, [_P, C.POINTER(_abi.GrayState), C.POINTER(_abi.Params),
                                                     C.POINTER(_abi.PrepareOpts)]),
    "rrtmgp_hip_update_fluxes": (C.c_int, [_P, C.POINTER(_abi.UpdateFluxesArgs)]),
    "rrtmgp_hip_update_fluxes_gray": (C.c_int, [_P, C.POINTER(_abi.UpdateFluxesGrayArgs)]),
    "rrtmgp_hip_workspace_transfer_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rrtmgp_hip_gas_lookup_create_multi": (C.c_int, [C.POINTER(_abi.GasLookupDesc), C.POINTER(C.c_int32), C.c_int,
                                                     C.POINTER(_P)]),
    "rrtmgp_hip_cloud_lookup_create_multi": (C.c_int, [C.POINTER(_abi.CloudLookupDesc), C.POINTER(C.c_int32), C.c_int,
                                                       C.POINTER(_P)]),
    "rrtmgp_hip_aerosol_lookup_create_multi": (C.c_int, [C.POINTER(_abi.AerosolLookupDesc), C.POINTER(C.c_int32), C.c_int,
                                                         C.POINTER(_P)]),
    "rrtmgp_hip_workspace_create_multi": (C.c_int, [C.POINTER(C.c_int32), C.c_int, C.c_int64, C.c_int64, C.c_int32,
                                                    C.POINTER(_P)]),
    "rrtmgp_hip_workspace_shards": (C.c_int, [_P]),
    "rrtmgp_hip_local_cpus": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.c_int]),
    "rrtmgp_hip_host_register": (C.c_int, [_P, C.c_size_t]),
    "rrtmgp_hip_host_unregister": (C.c_int, [_P]),
    "rrtmgp_hip_host_registered_count": (C.c_int, []),
    "rrtmgp_hip_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_P)]),
    "rrtmgp_hip_device_free": (C.c_int, [C.c_int, _P]),
    "rrtmgp_hip_memcpy": (C.c_int, [C.c_int, _P, _P, C.c_size_t, C.c_int32]),
    "rrtmgp_hip_memset": (C.c_int, [C.c_int, _P, C.c_int32, C.c_size_t]),
    "rrtmgp_hip_allocation_counts": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rrtmgp_hip_mcica_uniform": (C.c_double, [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "rrtmgp_hip_eval_primitive": (C.c_int, [C.c_int, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64]),
    "rrtmgp_hip_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "rrtmgp_hip_version": (C.c_char_p, []),
    "rrtmgp_hip_build_flags": (C.c_char_p, []),
    "rrtmgp_hip_abi_sizeof": (C.c_int, [C.c_int]),
}

ABI_STRUCTS = [_abi.MinorDesc, _abi.GasLookupDesc, _abi.CloudLookupDesc, _abi.AerosolLookupDesc, _abi.AtmosState,
               _abi.LwBcs, _abi.SwBcs, _abi.FluxOut, _abi.SolveOpts, _abi.GrayState, _abi.Params, _abi.PrepareOpts,
               _abi.View2D, _abi.UpdateFluxesArgs, _abi.UpdateFluxesGrayArgs]


class RRTMGPHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libhip_rrtmgp.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    for target in ([], ["fast"]):   # the shipped library (IEEE-accurate Float32), then the fast-forms build bench.py times next to it
        r = subprocess.run(["make", "-C", CSRC, "-j4"] + target, capture_output=True, text=True)
        if r.returncode != 0:
            raise RRTMGPHipError("building libhip_rrtmgp.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return os.path.join(_HERE, "libhip_rrtmgp.so")


def lib():
    """The loaded library with typed entry points; raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RRTMGPHipError(f"{SO_PATH} is missing: run `make -C {CSRC}` (or __graft_entry__.build()). "
                                 "There is no CPU fallback for the product path.")
        try:
            # When PyTorch is present it must be imported first: its wheel bundles a HIP runtime
            # with the same SONAME (libamdhip64.so.7), and the library has to share that one
            # runtime instance to use torch's streams and device tensors in place.
            # RRTMGP_HIP_NO_TORCH=1: a host-array caller that does not use torch tensors (what a Julia host is) runs on the
            # system's HIP runtime instead of the one bundled with the PyTorch wheel
            if not os.environ.get("RRTMGP_HIP_NO_TORCH"):
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            L = C.CDLL(SO_PATH)
        except OSError as e:
            raise RRTMGPHipError(f"cannot load {SO_PATH}: {e}") from e
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        for i, st in enumerate(ABI_STRUCTS):
            if L.rrtmgp_hip_abi_sizeof(i) != C.sizeof(st):
                raise RRTMGPHipError(f"ABI mismatch for {st.__name__}: library {L.rrtmgp_hip_abi_sizeof(i)} bytes, "
                                     f"binding {C.sizeof(st)} bytes")
        _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().rrtmgp_hip_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RRTMGPHipError(f"{what}: {_abi.ERRORS.get(rc, rc)}: {last_error()}")


def allocation_counts():
    """(hipMalloc calls, hipFree calls, hipHostRegister calls) the library has made since it was loaded."""
    a, f, r = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib().rrtmgp_hip_allocation_counts(C.byref(a), C.byref(f), C.byref(r)), "allocation_counts")
    return a.value, f.value, r.value


def require_gpu() -> int:
    n = lib().rrtmgp_hip_device_count()
    if n <= 0:
        raise RRTMGPHipError("no HIP device visible: " + last_error())
    return n


PRIMITIVES = {"exp_neg": 0, "exp_pair_e1": 1, "exp_pair_om1": 2, "rcp": 3, "div": 4, "sqrt_pos": 5, "ieee_div": 6}


def eval_primitive(name: str, x, y=None, device: int = 0, library=None):
    """One of the kernels' device math forms (include/rrtmgp_hip.h, rrtmgp_hip_eval_primitive) element-wise on numpy arrays.
    `library`: a ctypes handle of another build of the library (the IEEE-Float32 one); default: the loaded one."""
    import numpy as np
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.float32, np.float64)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=x.dtype)
        assert y.shape == x.shape
        yp = y.ctypes.data_as(C.c_void_p)
    L = library if library is not None else lib()
    fn = L.rrtmgp_hip_eval_primitive
    fn.restype, fn.argtypes = EXPORTS["rrtmgp_hip_eval_primitive"]
    rc = fn(device, PRIMITIVES[name], x.dtype.itemsize, x.ctypes.data_as(C.c_void_p), yp, out.ctypes.data_as(C.c_void_p), x.size)
    if rc != 0:
        raise RRTMGPHipError(f"eval_primitive({name}): {_abi.ERRORS.get(rc, rc)}")
    return out

