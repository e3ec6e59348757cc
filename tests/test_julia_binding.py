"""The Julia glue (ext/RRTMGPHIPExt.jl) cannot run here (no Julia in the image), but its
C-struct mirrors can be checked mechanically: same field names, order and sizes as the
ctypes mirror (which is itself checked against the compiled library in test_abi.py), and
every `ccall`ed symbol is declared in include/rrtmgp_hip.h."""
import ctypes as C
import os
import re

import pytest

from rrtmgp_jl_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "ext", "RRTMGPHIPExt.jl")).read()

PAIRS = {"MinorDesc": _abi.MinorDesc, "GasLookupDesc": _abi.GasLookupDesc, "CloudLookupDesc": _abi.CloudLookupDesc,
         "AerosolLookupDesc": _abi.AerosolLookupDesc, "AtmosStateDesc": _abi.AtmosState, "LwBcsDesc": _abi.LwBcs,
         "SwBcsDesc": _abi.SwBcs, "FluxOutDesc": _abi.FluxOut, "SolveOpts": _abi.SolveOpts,
         "GrayStateDesc": _abi.GrayState, "ParamsDesc": _abi.Params, "PrepareOpts": _abi.PrepareOpts,
         "View2D": _abi.View2D}
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


def julia_fields_of(name):
    body = re.search(r"^(?:mutable )?struct %s(?: <: [\w.]+)?\n(.*?)^end" % name, JL, flags=re.S | re.M).group(1)
    out = []
    for part in re.split(r"[;\n]", body):
        part = part.split("#")[0].strip()
        if "::" in part:
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


# ---- static checks of the method bodies and signatures (no Julia in the image) ----------------
import json
import sys

sys.path.insert(0, os.path.join(ROOT, "tools"))
import julia_lite as JLITE  # noqa: E402

SIGS = json.load(open(os.path.join(ROOT, "tests", "golden", "julia_signatures.json")))


def _module(src=JL):
    return JLITE.parse_module(src)


def _unresolved(src):
    mod = _module(src)
    names = mod.names()
    bad = []
    for m in mod.methods:
        for nm, line in JLITE.unresolved_names(m, names):
            bad.append((m.name, nm, line))
    return bad


def test_every_identifier_in_every_method_body_resolves():
    """Each name used in a body is a parameter, a local, a module-level name (struct, const, import, method)
    or a whitelisted Base name.  Round 1's glue used `band_flux` in three gray methods that do not take it."""
    mod = _module()
    assert len(mod.methods) >= 40 and len(mod.structs) >= 13, (len(mod.methods), len(mod.structs))
    assert _unresolved(JL) == []


def test_checker_catches_the_round1_defect():
    """The checker must FAIL on a method that names something it does not have (the round-1 `band_flux` bug)."""
    broken = JL.replace("flux_desc(flux_lw), opts()))\n    return nothing\nend\n\nfunction rte_lw_noscat_solve!(dev::HIPDevice, flux_lw::FluxLW",
                        "flux_desc(flux_lw, band_flux), opts()))\n    return nothing\nend\n\nfunction rte_lw_noscat_solve!(dev::HIPDevice, flux_lw::FluxLW", 1)
    assert broken != JL
    bad = _unresolved(broken)
    assert ("rte_lw_2stream_solve!", "band_flux") in {(m, n) for m, n, _ in bad}, bad


def _device_methods(mod):
    out = {}
    for m in mod.methods:
        if m.params and JLITE.norm_type(m.params[0].type) == "HIPDevice":
            out.setdefault(m.name.split(".")[-1], []).append(m)
    return out


def test_device_methods_have_the_reference_signatures():
    """For every device method of ext/cuda/*.jl (tests/golden/julia_signatures.json) the glue defines a method with
    the same name, the same number of positional parameters and, parameter by parameter, the same type annotation
    and default flag (the device type aside).  Catches arity drift and transposed arguments."""
    mine = _device_methods(_module())
    for ref in SIGS["methods"]:
        cands = mine.get(ref["name"], [])
        assert cands, f"no HIPDevice method {ref['name']} ({ref['file']}:{ref['line']})"
        want = [(p["type"], p["default"], p["vararg"]) for p in ref["params"][1:]]
        ok = False
        for m in cands:
            got = [(JLITE.norm_type(p.type), p.has_default, p.vararg) for p in m.params[1:]]
            if got == want:
                ok = True
                # same parameter NAMES too where the glue spells them out (forwarders use args...)
                names = [p.name for p in m.params[1:]]
                if not any(p.vararg for p in m.params):
                    assert names == [p["name"] for p in ref["params"][1:]], (ref["name"], names)
        assert ok, (ref["name"], ref["file"], ref["line"], want,
                    [[(JLITE.norm_type(p.type), p.has_default, p.vararg) for p in m.params[1:]] for m in cands])


def test_flux_layout_rule_matches_the_reference_allocation():
    """`_coalesced_2d` (src/optics/Fluxes.jl:45-49) dispatches on the ARRAY type: `Array` -> PermutedDimsArray over
    a (nlev, ncol) parent, anything else -> plain (ncol, nlev).  The glue must pick RRTMGP_LAYOUT_NLEV_NCOL for the
    first, RRTMGP_LAYOUT_NCOL_NLEV for the second, and hand over the PARENT's pointer of the wrapper."""
    header = open(os.path.join(ROOT, "include", "rrtmgp_hip.h")).read()
    c_layout = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (RRTMGP_LAYOUT_\w+)\s+(\d+)", header)}
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"const (LAYOUT_\w+) = Int32\((\d+)\)", JL)}
    assert consts == {"LAYOUT_NCOL_NLEV": c_layout["RRTMGP_LAYOUT_NCOL_NLEV"],
                      "LAYOUT_NLEV_NCOL": c_layout["RRTMGP_LAYOUT_NLEV_NCOL"]}
    rule = {}
    mod = _module()
    for m in mod.methods:
        if m.name == "flux_layout":
            rule[JLITE.norm_type(m.params[0].type)] = "".join(t.text for t in m.body if t.kind != "nl")
    perm = "PermutedDimsArray{T,2,(2,1),(2,1)}"
    assert rule == {perm: "LAYOUT_NLEV_NCOL", "AbstractMatrix": "LAYOUT_NCOL_NLEV"}, rule
    for entry in SIGS["coalesced_2d"]:
        if entry["array_type"] == "Type{Array}":      # what array_type(::HIPDevice) returns
            assert entry["wrapper"] == "PermutedDimsArray" and entry["parent_dims"] == ["d2", "d1"]   # (nlev, ncol) parent
        else:
            assert entry["wrapper"] == "none" and entry["parent_dims"] == ["d1", "d2"]                 # (ncol, nlev)
    assert re.search(r"ClimaComms\.array_type\(::HIPDevice\) = Array", JL)
    ptr_rule = {JLITE.norm_type(m.params[0].type): "".join(t.text for t in m.body if t.kind != "nl")
                for m in mod.methods if m.name == "flux_ptr"}
    assert ptr_rule == {perm: "ptr(parent(a))", "AbstractMatrix": "ptr(a)"}, ptr_rule
    # and both flux descriptors use the rule for every flux array
    for m in mod.methods:
        if m.name == "flux_desc":
            body = "".join(t.text for t in m.body if t.kind != "nl")
            assert "flux_layout(f.flux_up)" in body and "ptr(f.flux_" not in body.replace("flux_ptr(f.flux_", "")


def test_transposed_state_cache_is_switched_off_and_handles_are_released():
    assert re.search(r"RRTMGP\.RTE\._default_state_cache\(::HIPDevice, gp::RRTMGP\.RRTMGPGridParams\) = nothing", JL)
    assert "finalizer(release!, h)" in JL and "atexit(release_all!)" in JL
    for sym in ("rrtmgp_hip_workspace_destroy", "rrtmgp_hip_lookup_destroy", "rrtmgp_hip_workspace_create_multi"):
        assert f"(:{sym}, libhip[])" in JL, sym


def test_ccall_argument_counts_match_the_header():
    """Every ccall passes as many arguments as its type tuple declares, and as the C prototype takes."""
    header = open(os.path.join(ROOT, "include", "rrtmgp_hip.h")).read()
    toks = [t for t in JLITE.tokenize(JL) if t.kind != "nl"]
    n_checked = 0
    for i, t in enumerate(toks):
        if t.kind == "id" and t.text == "ccall" and toks[i + 1].text == "(":
            close = JLITE._matching(toks, i + 1)
            args = JLITE._split_top(toks[i + 2:close], ",")
            sym = args[0][1].text.lstrip(":")
            types = JLITE._split_top(args[2][1:-1], ",")
            assert len(args) - 3 == len(types), (sym, len(args) - 3, len(types))
            proto = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % sym, header, flags=re.S)
            assert proto, sym
            c_args = [a for a in proto.group(1).split(",") if a.strip() and a.strip() != "void"]
            assert len(c_args) == len(types), (sym, len(c_args), len(types))
            n_checked += 1
    assert n_checked >= 18, n_checked


# What the reference's constructors put into the fields that are typed by an unbounded type parameter
# (AtmosphericStates.jl:70-81, LookUpTables.jl:123-147, 174-205): the chains the glue follows through them.
FIELD_HINTS = {
    ("AtmosphericState", "cloud_state"): ["CloudState"],
    ("AtmosphericState", "aerosol_state"): ["AerosolState"],
    ("AtmosphericState", "vmr"): ["Vmr", "VmrGM"],
    ("LookUpLW", "planck"): ["LookUpPlanck"],
    ("LookUpLW", "band_data"): ["BandData"], ("LookUpSW", "band_data"): ["BandData"],
    ("LookUpLW", "ref_points"): ["ReferencePoints"], ("LookUpSW", "ref_points"): ["ReferencePoints"],
    ("LookUpLW", "minor_lower"): ["LookUpMinor"], ("LookUpLW", "minor_upper"): ["LookUpMinor"],
    ("LookUpSW", "minor_lower"): ["LookUpMinor"], ("LookUpSW", "minor_upper"): ["LookUpMinor"],
}


def _bad_fields(src):
    structs = SIGS["structs"]
    out = []
    for m in JLITE.parse_module(src).methods:
        out += [(m.name, chain, why) for chain, why, _ in JLITE.bad_field_accesses(m, structs, FIELD_HINTS)]
    return out


def test_every_field_access_names_a_field_of_the_reference_struct():
    """`as.layerdata`, `lkp.planck.tot_planck`, `cs.cld_frac` ...: every chain whose base is a parameter typed with a
    reference struct (or a local bound to such a chain) is followed through the reference's struct definitions
    (golden: names and annotations extracted by tools/julia_signatures.py).  A union-typed parameter must have the
    field in every member."""
    structs = SIGS["structs"]
    assert len(structs) >= 40 and [f for f, _ in structs["AtmosphericState"]["fields"]][:3] == ["lon", "lat", "layerdata"]
    mod = _module()
    typed = sum(1 for m in mod.methods for p in m.params if p.type and JLITE._type_structs(p.type, structs))
    assert typed >= 40, typed  # the check is not vacuous: that many parameters carry a reference struct type
    assert _bad_fields(JL) == []


def test_field_checker_catches_a_wrong_field_name():
    broken = JL.replace("ptr(cs.cld_frac)", "ptr(cs.cloud_frac)", 1)
    assert broken != JL
    assert ("state_desc", "cs.cloud_frac", "CloudState has no field cloud_frac") in _bad_fields(broken)
    broken = JL.replace("as.t_sfc", "as.t_surface", 1)
    assert any(why.endswith("has no field t_surface") for _, _, why in _bad_fields(broken))


def test_every_imported_and_qualified_reference_name_exists():
    """`import RRTMGP.Optics: compute_col_gas!`, `RP.grav(ps)`, `RRTMGP.RTE._default_state_cache(...)`: each name the
    glue takes from RRTMGP is defined in that module of the reference (golden index of names per module, built by
    tools/julia_signatures.py along the reference's include tree)."""
    mods = SIGS["modules"]
    assert {"RRTMGP", "RRTMGP.Optics", "RRTMGP.RTE", "RRTMGP.RTESolver", "RRTMGP.Parameters"} <= set(mods)
    code = "\n".join(l.split("#")[0] for l in JL.split("\n"))
    aliases, checked, missing = {}, 0, []
    for m in re.finditer(r"^import (RRTMGP(?:\.\w+)*) as (\w+)$", code, re.M):
        aliases[m.group(2)] = m.group(1)
    for m in re.finditer(r"^import (RRTMGP(?:\.\w+)*):((?:[^\n]|\n {4})*)", code, re.M):
        for nm in re.findall(r"[\w!]+", m.group(2)):
            checked += 1
            if nm not in mods.get(m.group(1), ()):
                missing.append(f"{m.group(1)}: {nm}")
    # qualified uses: RRTMGP.<Module>.<name>, RRTMGP.<name>, <alias>.<name>
    for m in re.finditer(r"\b(RRTMGP(?:\.[A-Z]\w*)*)\.([a-z_][\w!]*|[A-Z]\w*)\b", code):
        owner, nm = m.group(1), m.group(2)
        if owner + "." + nm in mods:   # a module path, not a name
            continue
        checked += 1
        if nm not in mods.get(owner, ()):
            missing.append(f"{owner}.{nm}")
    for al, owner in aliases.items():
        for m in re.finditer(r"\b" + al + r"\.([\w!]+)", code):
            checked += 1
            if m.group(1) not in mods[owner]:
                missing.append(f"{owner}.{m.group(1)}")
    assert checked >= 40, checked
    assert missing == []


# ---- the zero-allocation contract of the per-solve path (test/standalone.jl:361-383), held statically -----------------
def _hot(src=JL):
    mod = _module(src)
    ref_names = {r["name"] for r in SIGS["methods"]}
    roots = [m for ms in _device_methods(mod).values() for m in ms if m.name.split(".")[-1] in ref_names]
    assert len(roots) >= 13, [m.name for m in roots]
    return mod, JLITE.hot_methods(mod, roots)


def _allocations(src=JL):
    _, hot = _hot(src)
    return [(m.name, what, line) for m in hot for what, line in JLITE.allocating_constructs(m)]


def test_no_allocation_on_the_per_solve_path():
    """No function reachable from the 13 device methods broadcasts, copies a view into an `Array`, builds a `Dict` key,
    a closure, a `Ref` box or a string — except inside `*_slow` functions (handle creation on the first call, error
    text), which the walk does not enter.  The round-2 file failed this in five places (`Int32.(dev.ids)`
    4-5x per solve, `Array(view)` / `copyto!` around compute_col_gas! and compute_relative_humidity!, `get!(...) do`)."""
    mod, hot = _hot()
    names = {m.name.split(".")[-1] for m in hot}
    # the walk is not vacuous: it reaches the cache searches, the descriptors and the view conversion
    assert {"workspace", "lookup_handle", "state_desc", "flux_desc", "view2d", "opts", "check", "params_desc"} <= names, names
    assert not any(n.endswith("_slow") for n in names)
    assert _allocations() == []


@pytest.mark.parametrize("old,new,what", [
    # round 2's `devices(dev)`: a fresh Vector per call
    ("    c = dev.cache\n    ft = ftype(FT)\n", "    c = dev.cache\n    ft = ftype(FT)\n    ids32 = Int32.(dev.ids)\n", "broadcast `Int32.(...)`"),
    # round 2's dense copies of the strided arguments
    ("view2d(a::StridedMatrix) = View2D(ptr(a), stride(a, 1), stride(a, 2))",
     "view2d(a::StridedMatrix) = View2D(ptr(Array(a)), stride(a, 1), stride(a, 2))", "call of `Array`"),
    # round 2's `get!(WORKSPACES, (dev.ids, ...)) do ... end`
    ("    return workspace_create_slow(dev, Int(ncol), Int(nlay), ft)\n",
     "    return get!(() -> workspace_create_slow(dev, Int(ncol), Int(nlay), ft), TABLE, (dev.ids, ncol, nlay, ft))\n", "call of `get!`"),
    # an error message built on the hot path
    ("check(rc::Cint) = rc == 0 ? nothing : fail_slow(rc)", 'check(rc::Cint) = rc == 0 ? nothing : error("status $rc")', "call of `error`"),
])
def test_allocation_lint_catches_the_round2_patterns(old, new, what):
    assert old in JL, old
    found = {w for _, w, _ in _allocations(JL.replace(old, new, 1))}
    assert what in found, found


def test_strided_views_cross_the_abi_uncopied():
    """compute_col_gas! / compute_relative_humidity! / compute_gray_heating_rate! hand the reference's views over as
    (pointer, strides): `view2d` is defined on StridedMatrix, every 2-D array argument of the three ccalls goes through it,
    and the extents travel with the call."""
    mod = _module()
    v = [m for m in mod.methods if m.name == "view2d"]
    assert {JLITE.norm_type(m.params[0].type) for m in v} == {"StridedMatrix", "Nothing"}
    body = {m.name.split(".")[-1]: "".join(t.text for t in m.body if t.kind != "nl") for m in mod.methods
            if m.name.split(".")[-1] in ("compute_col_gas!", "compute_relative_humidity!", "compute_gray_heating_rate!")
            and m.params and JLITE.norm_type(m.params[0].type) == "HIPDevice"}
    assert body["compute_col_gas!"].count("view2d(") == 3 and body["compute_relative_humidity!"].count("view2d(") == 4
    assert body["compute_gray_heating_rate!"].count("view2d(") == 3
    for b in body.values():
        assert "ncol,nlay,view2d(" in b and "Array(" not in b and "copyto!" not in b


def test_device_owns_its_handles():
    """`HIPDevice` converts its ids to the ABI's Int32 once and owns a HandleCache; lookups are cached per device set
    (one global table keyed by the lookup alone served the wrong replica set to a second device)."""
    fields = dict(julia_fields_of("HIPDevice"))
    assert fields == {"ids": "Vector{Int32}", "cache": "HandleCache"}, fields
    assert "const LOOKUPS" not in JL and "const WORKSPACES" not in JL
    assert re.search(r"lk_table::Vector\{Any\}", JL)        # keeps the key array alive: its address cannot be reused


def test_allocation_lint_fails_on_the_round2_file():
    """tests/golden/RRTMGPHIPExt_round2.jl is this repository's own extension as round 2 left it (git ecb1d6f).  The lint
    must find what the review found there: a fresh `Int32.(dev.ids)` in `devices`, called from every lookup / workspace
    search of every solve; `Array(view)` + `copyto!` around the three view-taking methods; `get!(...) do` with a freshly
    built tuple key."""
    old = open(os.path.join(ROOT, "tests", "golden", "RRTMGPHIPExt_round2.jl")).read()
    found = _allocations(old)
    by_fn = {}
    for fn, what, _ in found:
        by_fn.setdefault(fn.split(".")[-1], set()).add(what)
    assert "broadcast `Int32.(...)`" in by_fn.get("devices", ()), by_fn
    for fn in ("compute_col_gas!", "compute_relative_humidity!", "compute_gray_heating_rate!"):
        assert "call of `Array`" in by_fn.get(fn, ()) and "call of `copyto!`" in by_fn.get(fn, ()), (fn, by_fn.get(fn))
    assert "call of `get!`" in by_fn.get("workspace", ()) and "closure" in by_fn.get("workspace", ()), by_fn.get("workspace")
    assert "call of `get!`" in by_fn.get("lookup_handle", ()), by_fn.get("lookup_handle")
    assert len(found) >= 20, len(found)


# ---- ccall argument TYPES against the C prototypes -------------------------------------------------------------------
_C2JL = {
    "int": {"Cint"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64"}, "uint64_t": {"UInt64"}, "double": {"Float64"},
    "size_t": {"Csize_t"}, "char*": {"Ptr{UInt8}"}, "constint32_t*": {"Ptr{Int32}"},
    "void*": {"P", "Ptr{Cvoid}"}, "constvoid*": {"P", "Ptr{Cvoid}"},
    "rrtmgp_workspace*": {"P", "Ptr{Cvoid}"}, "constrrtmgp_workspace*": {"P", "Ptr{Cvoid}"},
    "rrtmgp_lookup*": {"P", "Ptr{Cvoid}"}, "constrrtmgp_lookup*": {"P", "Ptr{Cvoid}"},
    "rrtmgp_lookup**": {"Ref{Ptr{Cvoid}}"}, "rrtmgp_workspace**": {"Ref{Ptr{Cvoid}}"},
}
_STRUCT2JL = {"rrtmgp_gas_lookup_desc": "GasLookupDesc", "rrtmgp_cloud_lookup_desc": "CloudLookupDesc",
              "rrtmgp_aerosol_lookup_desc": "AerosolLookupDesc", "rrtmgp_atmos_state": "AtmosStateDesc",
              "rrtmgp_lw_bcs": "LwBcsDesc", "rrtmgp_sw_bcs": "SwBcsDesc", "rrtmgp_flux_out": "FluxOutDesc",
              "rrtmgp_solve_opts": "SolveOpts", "rrtmgp_gray_state": "GrayStateDesc", "rrtmgp_params": "ParamsDesc",
              "rrtmgp_prepare_opts": "PrepareOpts", "rrtmgp_view2d": "View2D"}


def test_ccall_argument_types_match_the_header():
    """Argument by argument: an `int64_t` extent must be passed as Int64 (not Int32 / Cint), a `const rrtmgp_view2d *` as
    Ref{View2D}, handles as pointers, the return type as Cint (or Cstring / Cdouble where the header says so).  Catches the
    class of ABI bug a count check cannot: a 32-bit integer in a 64-bit slot."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rrtmgp_hip.h")).read(), flags=re.S)
    toks = [t for t in JLITE.tokenize(JL) if t.kind != "nl"]
    n_checked = 0
    for i, t in enumerate(toks):
        if not (t.kind == "id" and t.text == "ccall" and toks[i + 1].text == "("):
            continue
        close = JLITE._matching(toks, i + 1)
        args = JLITE._split_top(toks[i + 2:close], ",")
        sym = args[0][1].text.lstrip(":")
        ret = "".join(x.text for x in args[1])
        types = ["".join(x.text for x in a) for a in JLITE._split_top(args[2][1:-1], ",")]
        proto = re.search(r"([\w \*]+?)\b%s\s*\(([^;]*?)\)\s*;" % sym, header, flags=re.S)
        assert proto, sym
        c_ret = proto.group(1).split()[-1] if proto.group(1).split() else "int"
        assert ret == {"int": "Cint", "double": "Cdouble"}.get(c_ret, ret), (sym, ret, c_ret)
        c_args = [a.strip() for a in proto.group(2).split(",") if a.strip() and a.strip() != "void"]
        assert len(c_args) == len(types), sym
        for k, (ca, jt) in enumerate(zip(c_args, types)):
            words = ca.replace("*", " * ").split()
            if words[-1] != "*" and not words[-1].endswith("_t") and words[-1] not in ("int", "double", "size_t"):
                words = words[:-1]                       # drop the parameter name
            ctype = "".join(words)
            base = ctype.replace("const", "").rstrip("*")
            if base in _STRUCT2JL and ctype.endswith("*") and not ctype.endswith("**"):
                allowed = {"Ref{%s}" % _STRUCT2JL[base]}
            else:
                allowed = _C2JL.get(ctype)
            assert allowed is not None, (sym, k, ca, ctype)
            assert jt in allowed, f"{sym}: argument {k + 1} is `{ca}` in the header but `{jt}` in the ccall (allowed: {sorted(allowed)})"
            n_checked += 1
    assert n_checked >= 100, n_checked


def test_every_array_owner_used_in_a_ccall_is_gc_preserved():
    """`ptr(x)` / `pointer(x)` hand raw addresses to C: the object they came from must be rooted for the duration of the
    call (`GC.@preserve`).  For every device method, every parameter that carries arrays (by its annotation) and is named
    inside the ccall's argument list must also be named in the `GC.@preserve` that wraps that ccall."""
    carriers = ("AbstractArray", "AtmosphericState", "GrayAtmosphericState", "LwBCs", "SwBCs", "FluxLW", "FluxSW")
    mod = _module()
    n = 0
    for ms in _device_methods(mod).values():
        for m in ms:
            toks = [t for t in m.body if t.kind != "nl"]
            idx = [i for i, t in enumerate(toks) if t.kind == "id" and t.text == "ccall"]
            if not idx:
                continue
            params = {p.name: p.type for p in m.params if p.name}
            # untyped parameters of the gray heating-rate method are arrays too (reference signature has no annotations)
            arrays = {nm for nm, ty in params.items() if any(c in ty for c in carriers)}
            if m.name.split(".")[-1] == "compute_gray_heating_rate!":
                arrays |= {"hr_lay", "p_lev", "flux_net"}
            for i in idx:
                close = JLITE._matching(toks, i + 1)
                used = {t.text for t in toks[i + 2:close] if t.kind == "id"} & arrays
                # the @preserve list: identifiers between the macro and the `check(` that wraps the ccall
                j = max(k for k in range(i) if toks[k].kind == "macro" and toks[k].text.endswith("@preserve"))
                k = next(q for q in range(j + 1, i) if toks[q].text == "check")
                preserved = {t.text for t in toks[j + 1:k] if t.kind == "id"}
                missing = used - preserved
                assert not missing, (m.name, sorted(missing), sorted(preserved))
                n += len(used)
    assert n >= 30, n


def test_descriptor_constructors_pass_one_value_per_field():
    """The C-struct mirrors are built with positional constructors: `AtmosStateDesc(0, kind, ncol, ...)`.  One argument too
    few or too many is a MethodError at the first solve; a splatted tuple (`band_ptrs(band)...`) counts as the length of
    the tuple its methods return."""
    toks = [t for t in JLITE.tokenize(JL) if t.kind != "nl"]
    mod = _module()
    tuple_len = {}
    for m in mod.methods:       # short methods returning a tuple literal: name -> its length (all methods must agree)
        body = [t for t in m.body if t.kind != "nl"]
        if body and body[0].text == "(" and JLITE._matching(body, 0) == len(body) - 1:
            n = len(JLITE._split_top(body[1:-1], ","))
            tuple_len.setdefault(m.name, set()).add(n)
    assert tuple_len.get("band_ptrs") == {4}, tuple_len.get("band_ptrs")
    n_calls = 0
    for i, t in enumerate(toks):
        if t.kind == "id" and t.text in PAIRS and i + 1 < len(toks) and toks[i + 1].text == "(" and toks[i - 1].text not in ("struct", "{", "::"):
            close = JLITE._matching(toks, i + 1)
            args = JLITE._split_top(toks[i + 2:close], ",")
            n = 0
            for a in args:
                if a and a[-1].text == "..." and a[0].kind == "id" and a[0].text in tuple_len:
                    (k,) = tuple_len[a[0].text]
                    n += k
                else:
                    n += 1
            want = len(julia_fields(t.text))
            assert n == want, f"{t.text}(...) at line {t.line}: {n} arguments for {want} fields"
            n_calls += 1
    assert n_calls >= 20, n_calls


def test_descriptor_arguments_are_not_transposed():
    """Positional constructors again: when an argument is `ptr(x.name)` (or `flux_ptr(f.name)`, `pointer(l.name)`) and
    `name` is also the name of a field of the C struct, it must sit at THAT field's position — `ptr(as.t_lev)` in the
    `p_lev` slot compiles and runs, and gives wrong fluxes."""
    toks = [t for t in JLITE.tokenize(JL) if t.kind != "nl"]
    checked = 0
    for i, t in enumerate(toks):
        if not (t.kind == "id" and t.text in PAIRS and toks[i + 1].text == "(" and toks[i - 1].text not in ("struct", "{", "::")):
            continue
        names = [f for f, _ in julia_fields(t.text)]
        close = JLITE._matching(toks, i + 1)
        pos = 0
        for a in JLITE._split_top(toks[i + 2:close], ","):
            if a and a[-1].text == "...":
                pos += 4            # band_ptrs(band)... (checked by the arity test)
                continue
            # the last `.field` inside the argument, ignoring ternaries' C_NULL branches
            fields = []
            for k in range(len(a) - 1):
                if a[k].kind == "id" and a[k].text in ("ptr", "pointer", "flux_ptr") and a[k + 1].text == "(":
                    inner = a[k + 2:JLITE._matching(a, k + 1)]
                    fields += [inner[q + 1].text for q in range(len(inner) - 1) if inner[q].text == "." and inner[q + 1].kind == "id"][-1:]
            if fields and fields[-1] in names:
                assert names[pos] == fields[-1], (f"{t.text}(...) line {t.line}: argument {pos + 1} reads `.{fields[-1]}` "
                                                   f"but fills the field `{names[pos]}`")
                checked += 1
            pos += 1
    assert checked >= 45, checked
