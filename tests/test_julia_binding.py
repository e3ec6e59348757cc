"""The Julia glue (ext/RRTMGPHIPExt.jl) cannot run here (no Julia in the image), but its
C-struct mirrors can be checked mechanically: same field names, order and sizes as the
ctypes mirror (which is itself checked against the compiled library in test_abi.py), and
every `ccall`ed symbol is declared in include/rrtmgp_hip.h."""
import ctypes as C
import os
import re

from rrtmgp_jl_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "ext", "RRTMGPHIPExt.jl")).read()

PAIRS = {"MinorDesc": _abi.MinorDesc, "GasLookupDesc": _abi.GasLookupDesc, "CloudLookupDesc": _abi.CloudLookupDesc,
         "AerosolLookupDesc": _abi.AerosolLookupDesc, "AtmosStateDesc": _abi.AtmosState, "LwBcsDesc": _abi.LwBcs,
         "SwBcsDesc": _abi.SwBcs, "FluxOutDesc": _abi.FluxOut, "SolveOpts": _abi.SolveOpts,
         "GrayStateDesc": _abi.GrayState, "ParamsDesc": _abi.Params, "PrepareOpts": _abi.PrepareOpts}
SIZES = {"Int32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "P": 8, "Ptr{Int64}": 8, "MinorDesc": C.sizeof(_abi.MinorDesc),
         "NTuple{5, Float64}": 40}


def julia_fields(name):
    body = re.search(r"^struct %s\n(.*?)^end" % name, JL, flags=re.S | re.M).group(1)
    out = []
    for part in re.split(r"[;\n]", body):
        part = part.strip()
        if part:
            f, t = part.split("::")
            out.append((f.strip(), t.strip()))
    return out


def test_julia_struct_mirrors_match_ctypes():
    for jl_name, ct in PAIRS.items():
        jf = julia_fields(jl_name)
        cf = list(ct._fields_)
        assert [f for f, _ in jf] == [f for f, _ in cf], jl_name
        for (f, jt), (_, cty) in zip(jf, cf):
            assert SIZES[jt] == C.sizeof(cty), (jl_name, f, jt)


def test_julia_abi_struct_order_matches_library_numbering():
    order = re.search(r"const ABI_STRUCTS = \((.*?)\)", JL, flags=re.S).group(1)
    names = [n.strip() for n in order.replace("\n", " ").split(",") if n.strip()]
    assert [PAIRS[n] for n in names] == _lib.ABI_STRUCTS


def test_every_ccall_symbol_is_declared():
    header = open(os.path.join(ROOT, "include", "rrtmgp_hip.h")).read()
    syms = set(re.findall(r"\(:(rrtmgp_hip_[a-z0-9_]+), libhip\[\]\)", JL))
    assert len(syms) >= 14
    for s in syms:
        assert re.search(r"\b%s\s*\(" % s, header), s
