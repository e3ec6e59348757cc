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
         "View2D": _abi.View2D, "UpdateFluxesArgs": _abi.UpdateFluxesArgs,
         "UpdateFluxesGrayArgs": _abi.UpdateFluxesGrayArgs}
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
        # (`as` is a Python keyword: the ctypes mirror spells that one field `as_`)
        assert [f for f, _ in jf] == [f[:-1] if f == "as_" else f for f, _ in cf], jl_name
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
    broken = JL.replace("set!(sc.flux_lw, flux_desc(flux_lw)), set!(sc.opts, opts())))\n    return nothing\nend\n\nfunction rte_lw_noscat_solve!(dev::HIPDevice, flux_lw::FluxLW",
                        "set!(sc.flux_lw, flux_desc(flux_lw, band_flux)), set!(sc.opts, opts())))\n    return nothing\nend\n\nfunction rte_lw_noscat_solve!(dev::HIPDevice, flux_lw::FluxLW", 1)
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
    # host arrays by default; `resident = true` switches to the device-resident type, whose flux buffers are plain
    # (ncol, nlev) arrays by the same rule (`_coalesced_2d` dispatches on the ARRAY type: anything but Array)
    assert re.search(r"ClimaComms\.array_type\(dev::HIPDevice\) = dev\.resident \? HIPArray : Array", JL)
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


# the Layer-2 overrides reach the solver's parts through fields typed by unbounded parameters (src/api/solver.jl:95-134,
# src/rte/RTE.jl:53-287): what the RRTMGPSolver constructor puts there for spectral radiation
FIELD_HINTS.update({
    ("RRTMGPSolver", "grid_params"): ["RRTMGPGridParams"], ("RRTMGPSolver", "as"): ["AtmosphericState"],
    ("RRTMGPSolver", "lws"): ["TwoStreamLWRTE", "NoScatLWRTE"], ("RRTMGPSolver", "sws"): ["TwoStreamSWRTE"],
    ("RRTMGPSolver", "lookups"): ["LookupBundle"], ("RRTMGPSolver", "radiation_method"): ["AllSkyRadiation"],
    ("RRTMGPSolver", "clear_flux_lw"): ["FluxPresentation"], ("RRTMGPSolver", "clear_flux_sw"): ["FluxPresentation"],
    ("TwoStreamLWRTE", "bcs"): ["LwBCs"], ("NoScatLWRTE", "bcs"): ["LwBCs"], ("TwoStreamSWRTE", "bcs"): ["SwBCs"],
    ("NoScatLWRTE", "angle_disc"): ["AngularDiscretization"],
    ("TwoStreamLWRTE", "band_flux"): ["FluxBand"], ("TwoStreamSWRTE", "band_flux"): ["FluxBand"],
    ("LookupBundle", "lookup_lw"): ["LookUpLW"], ("LookupBundle", "lookup_sw"): ["LookUpSW"],
})
TYPE_ALIASES = {"HIPSpectralSolver": ["RRTMGPSolver"], "HIPGraySolver": ["RRTMGPSolver"]}


def _bad_fields(src):
    structs = SIGS["structs"]
    out = []
    for m in JLITE.parse_module(src).methods:
        out += [(m.name, chain, why) for chain, why, _ in JLITE.bad_field_accesses(m, structs, FIELD_HINTS, TYPE_ALIASES)]
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


def test_field_checker_follows_the_solver_through_the_type_alias():
    """`s::HIPSpectralSolver` is an RRTMGPSolver: `s.lws.bcs.sfc_emis`, `s.presented_flux_lw.flux_net`, ... are checked too."""
    broken = JL.replace("set!(sc.bcs_lw, lw_bcs_desc(lws.bcs))\n    set!(sc.bcs_sw, sw_bcs_desc(sws.bcs))",
                        "set!(sc.bcs_lw, lw_bcs_desc(s.lws.bcs))\n    set!(sc.bcs_sw, sw_bcs_desc(s.sws.boundary))", 1)
    assert broken != JL
    assert any(chain == "s.sws.boundary" for _, chain, _ in _bad_fields(broken)), _bad_fields(broken)
    broken = JL.replace("s.presented_flux_sw.flux_net)", "s.presented_flux_sw.net)", 1)
    assert broken != JL and any(why == "FluxPresentation has no field net" for _, _, why in _bad_fields(broken))
    broken = JL.replace("ptr(s.net_flux_buffer)", "ptr(s.net_flux)", 1)
    assert broken != JL and any(why == "RRTMGPSolver has no field net_flux" for _, _, why in _bad_fields(broken))


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
    for m in re.finditer(r"\b(RRTMGP(?:\.[A-Z]\w*)*)\.([a-z_][\w!]*|[A-Z]\w*)(?![\w!])", code):
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


L2_NAMES = ("update_fluxes!", "update_lw_fluxes!", "update_sw_fluxes!", "update_net_fluxes!", "prepare_atmosphere!")


def _layer2_methods(mod):
    return [m for m in mod.methods if m.name.split(".")[-1] in L2_NAMES and m.params
            and JLITE.norm_type(m.params[0].type) == "HIPSpectralSolver"]


def test_layer2_overrides_have_the_reference_signatures():
    """update_fluxes!(s, seedval = nothing), prepare_atmosphere!(s), update_lw_fluxes!(s), update_sw_fluxes!(s),
    update_net_fluxes!(s) (src/api/update_fluxes.jl:12,74,165,223,252; golden index `l2_methods`): the glue specialises each
    of them on the solver type alone — same name, same positional parameters, same defaults — for solvers whose grid
    parameters carry a HIPDevice context and whose radiation method is spectral."""
    mine = {}
    for m in _layer2_methods(_module()):
        mine.setdefault(m.name.split(".")[-1], []).append(m)
    assert set(mine) == set(L2_NAMES), sorted(mine)
    for name in L2_NAMES:
        # the reference's entry method of that name: the one that takes the solver and no radiation-method tag
        refs = [r for r in SIGS["l2_methods"] if r["name"] == name and all(p["name"] for p in r["params"])]
        assert len(refs) == 1, (name, refs)
        want = [(p["name"], p["default"]) for p in refs[0]["params"]]
        (m,) = mine[name]
        assert [(p.name, p.has_default) for p in m.params] == want, (name, want)
        assert refs[0]["params"][0]["type"] == "RRTMGPSolver"
    # the dispatch chain: solver -> grid params -> context -> device, and spectral methods only (gray keeps the generic path)
    for pat in (r"const HIPContext = ClimaComms\.SingletonCommsContext\{<:HIPDevice\}",
                r"const HIPGrid = RRTMGP\.RRTMGPGridParams\{<:Any, <:HIPContext\}",
                r"const SpectralMethod = Union\{RRTMGP\.ClearSkyRadiation, RRTMGP\.AllSkyRadiation, RRTMGP\.AllSkyRadiationWithClearSkyDiagnostics\}",
                r"const HIPSpectralSolver = RRTMGP\.RRTMGPSolver\{<:HIPGrid, <:SpectralMethod\}"):
        assert re.search(pat, JL), pat
    g = SIGS["structs"]["RRTMGPGridParams"]
    assert list(g["params"]) == ["FT", "C"] and [f for f, _ in g["fields"]][0] == "context"     # {FT, C}: C is the context type
    sp = list(SIGS["structs"]["RRTMGPSolver"]["params"])
    assert sp[:2] == ["S", "RM"]                                                               # {S = grid params, RM = method, ...}
    assert dict(SIGS["structs"]["RRTMGPSolver"]["fields"])["grid_params"] == "S"
    assert dict(SIGS["structs"]["RRTMGPSolver"]["fields"])["radiation_method"] == "RM"


def test_layer2_step_is_one_library_call_with_the_state_staged_once():
    """update_fluxes!(s::HIPSpectralSolver) makes exactly one ccall — rrtmgp_hip_update_fluxes — keeps the reference's
    validation and seeding steps (update_fluxes.jl:226-227), writes the presentation arrays the getters read, and hands
    the nested descriptors over as addresses of the device's scratch slots (rooted by GC.@preserve)."""
    (m,) = [x for x in _layer2_methods(_module()) if x.name.split(".")[-1] == "update_fluxes!"]
    body = "".join(t.text for t in m.body if t.kind != "nl")
    assert body.count("ccall(") == 1 and "(:rrtmgp_hip_update_fluxes,libhip[])" in body
    assert "RRTMGP.check_values[]&&RRTMGP.validate_inputs(s)" in body and "RRTMGP._maybe_reset_rng_seed!(s.radiation_method,seedval)" in body
    for need in ("presented_desc(s.presented_flux_lw,lw_band(lws),s.clear_flux_lw)", "presented_desc(s.presented_flux_sw,sws.band_flux,s.clear_flux_sw)",
                 "ptr(s.net_flux_buffer),ptr(s.clear_net_flux_buffer)", "GC.@preservessc", "refptr(sc.prepare)"):
        assert need in body, need
    # ... and the separately callable steps never copy through the solvers' compute buffers
    for x in _layer2_methods(_module()):
        b = "".join(t.text for t in x.body if t.kind != "nl")
        assert "update_presentation!" not in b and ".lws.flux" not in b and ".sws.flux" not in b, x.name


# ---- the zero-allocation contract of the per-solve path (test/standalone.jl:361-383), held statically -----------------
def _hot(src=JL):
    mod = _module(src)
    ref_names = {r["name"] for r in SIGS["methods"]}
    roots = [m for ms in _device_methods(mod).values() for m in ms if m.name.split(".")[-1] in ref_names]
    assert len(roots) >= 13, [m.name for m in roots]
    roots += _layer2_methods(mod)       # the Layer-2 overrides are per-step code too (none in the round-2 fixture)
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
    # ... and everything under the Layer-2 overrides
    assert {"update_fluxes!", "update_lw_fluxes!", "update_sw_fluxes!", "update_net_fluxes!", "prepare_atmosphere!", "lw_solve!",
            "sw_solve!", "presented_desc", "step_opts", "step_prepare", "prepare_desc", "add_into!", "set!", "refptr",
            "cloud_lookups", "step_workspace"} <= names, names
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
    assert {JLITE.norm_type(m.params[0].type) for m in v} == {"StridedMatrix", "Nothing", "AbstractMatrix"}
    # anything without a (pointer, strides) form is refused with a message (a cold `_slow` function), never copied
    fallback = next(m for m in v if JLITE.norm_type(m.params[0].type) == "AbstractMatrix")
    assert "".join(t.text for t in fallback.body if t.kind != "nl") == "not_strided_slow(a)"
    body = {m.name.split(".")[-1]: "".join(t.text for t in m.body if t.kind != "nl") for m in mod.methods
            if m.name.split(".")[-1] in ("compute_col_gas!", "compute_relative_humidity!", "compute_gray_heating_rate!")
            and m.params and JLITE.norm_type(m.params[0].type) == "HIPDevice"}
    assert body["compute_col_gas!"].count("view2d(") == 3 and body["compute_relative_humidity!"].count("view2d(") == 4
    assert body["compute_gray_heating_rate!"].count("view2d(") == 3
    for b in body.values():
        assert "ncol,nlay,set!(sc.v1,view2d(" in b and "Array(" not in b and "copyto!" not in b


def test_device_owns_its_handles():
    """`HIPDevice` converts its ids to the ABI's Int32 once and owns a HandleCache; lookups are cached per device set
    (one global table keyed by the lookup alone served the wrong replica set to a second device)."""
    fields = dict(julia_fields_of("HIPDevice"))
    assert fields == {"ids": "Vector{Int32}", "cache": "HandleCache", "resident": "Bool"}, fields
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
    "rrtmgp_lookup**": {"Ref{Ptr{Cvoid}}"}, "rrtmgp_workspace**": {"Ref{Ptr{Cvoid}}"}, "void**": {"Ref{Ptr{Cvoid}}"},
}
_STRUCT2JL = {"rrtmgp_gas_lookup_desc": "GasLookupDesc", "rrtmgp_cloud_lookup_desc": "CloudLookupDesc",
              "rrtmgp_aerosol_lookup_desc": "AerosolLookupDesc", "rrtmgp_atmos_state": "AtmosStateDesc",
              "rrtmgp_lw_bcs": "LwBcsDesc", "rrtmgp_sw_bcs": "SwBcsDesc", "rrtmgp_flux_out": "FluxOutDesc",
              "rrtmgp_solve_opts": "SolveOpts", "rrtmgp_gray_state": "GrayStateDesc", "rrtmgp_params": "ParamsDesc",
              "rrtmgp_prepare_opts": "PrepareOpts", "rrtmgp_view2d": "View2D", "rrtmgp_update_fluxes_args": "UpdateFluxesArgs",
              "rrtmgp_update_fluxes_gray_args": "UpdateFluxesGrayArgs"}


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
    carriers = ("AbstractArray", "AtmosphericState", "GrayAtmosphericState", "LwBCs", "SwBCs", "FluxLW", "FluxSW",
                "HIPSpectralSolver")
    mod = _module()
    n = 0
    for ms in list(_device_methods(mod).values()) + [_layer2_methods(mod)]:
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
    assert n_calls >= 18, n_calls


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
                if a[k].kind == "id" and a[k].text in ("ptr", "pointer", "flux_ptr", "refptr") and a[k + 1].text == "(":
                    inner = a[k + 2:JLITE._matching(a, k + 1)]
                    fields += [inner[q + 1].text for q in range(len(inner) - 1) if inner[q].text == "." and inner[q + 1].kind == "id"][-1:]
            if fields and fields[-1] in names:
                assert names[pos] == fields[-1], (f"{t.text}(...) line {t.line}: argument {pos + 1} reads `.{fields[-1]}` "
                                                   f"but fills the field `{names[pos]}`")
                checked += 1
            pos += 1
    assert checked >= 45, checked


# ---- what a Julia user needs to LOAD the extension: trigger package, Project patch, usage example --------------------
def test_trigger_package_patch_and_example_are_consistent():
    """julia/HIPRRTMGP (the weak dependency that triggers RRTMGPHIPExt, like CUDA triggers RRTMGPCUDAExt: reference
    Project.toml:14-22), julia/Project.toml.patch and examples/julia_usage.jl: one uuid everywhere, the patch adds exactly
    the weak dependency and the extension entry, and every name the example takes from the extension or from RRTMGP exists."""
    proj = open(os.path.join(ROOT, "julia", "HIPRRTMGP", "Project.toml")).read()
    uuid = re.search(r'^uuid = "([0-9a-f-]{36})"', proj, re.M).group(1)
    assert re.search(r'^name = "HIPRRTMGP"', proj, re.M) and "<uuid>" not in proj
    patch = open(os.path.join(ROOT, "julia", "Project.toml.patch")).read()
    added = [ln[1:] for ln in patch.splitlines() if ln.startswith("+") and not ln.startswith("+++")]
    assert added == [f'HIPRRTMGP = "{uuid}"', 'RRTMGPHIPExt = "HIPRRTMGP"'], added
    assert not [ln for ln in patch.splitlines() if ln.startswith("-") and not ln.startswith("---")]
    pkg = open(os.path.join(ROOT, "julia", "HIPRRTMGP", "src", "HIPRRTMGP.jl")).read()
    assert 'Base.UUID("a01a1ee8-cea4-48fc-987c-fc7878d79da1"), "RRTMGP"' in pkg      # RRTMGP.jl's uuid (reference Project.toml:2)
    assert "Base.get_extension(rr, :RRTMGPHIPExt)" in pkg and "const libpath" in pkg and 'ENV["RRTMGP_HIP_LIBRARY"]' in pkg
    assert 'libhip[] = get(ENV, "RRTMGP_HIP_LIBRARY", libhip[])' in JL                # ... which the extension reads at load time
    pm = JLITE.parse_module(pkg)
    assert {m.name for m in pm.methods} >= {"extension", "__init__"} and "libpath" in pm.consts
    ext_names = _module().names() | set(_module().exports)
    for script in (os.path.join(ROOT, "examples", "julia_usage.jl"), os.path.join(ROOT, "julia", "test_hip_extension.jl")):
        src = "\n".join(ln.split("#")[0] for ln in open(script).read().split("\n"))
        toks = JLITE.tokenize(src)                                  # it tokenizes: balanced strings, no stray characters
        assert sum(t.text == "(" for t in toks) == sum(t.text == ")" for t in toks)
        for nm in re.findall(r"\bHIP\.([\w!]+)", src):
            assert nm in ext_names, (script, nm)
        for nm in re.findall(r"\bRRTMGP\.([\w!]+)", src):
            assert nm in SIGS["modules"]["RRTMGP"], (script, nm)
    ex = open(os.path.join(ROOT, "examples", "julia_usage.jl")).read()
    for need in ("HIPRRTMGP.extension()", "HIP.HIPDevice(0)", "ClimaComms.SingletonCommsContext(device)", "RRTMGP.RRTMGPGridParams(FT; context",
                 "HIP.pin!(solver)", "RRTMGP.update_fluxes!(solver)", "RRTMGP.net_flux(solver)"):
        assert need in ex, need


def test_pin_walk_is_bounded_and_unpin_parks_refused_arrays():
    """`pin!` recursed through struct fields without a visited set (a cyclic object graph overflowed the stack)
    and dropped the status of host_register / host_unregister.  Now: an IdSet of visited objects and a depth cap, a warning
    on a refused registration, and an array whose unregistration the library refuses (a solve is still using the range)
    is parked and retried instead of being forgotten."""
    mod = _module()
    by = {m.name: "".join(t.text for t in m.body if t.kind != "nl") for m in mod.methods}
    assert "Base.IdSet{Any}()" in by["pin!"] and "max_depth" in by["pin!"]
    w = by["pin_walk_slow"]
    assert "xinseen||depth<0" in w and "push!(seen,x)" in w and "depth-1" in w and "@warn" in w and "finalizer(unpin_slow,x)" in w
    assert "rc==0||push!(PARKED,a)" in by["unpin_slow"] and "retry_parked_slow()" in by["unpin_slow"]
    assert "atexit(retry_parked_slow)" in by["__init__"]


def test_device_resident_array_type_is_wired_through_every_descriptor():
    """N4 (round 5): `HIPDevice(id; resident = true)` makes `array_type` a device-resident `HIPArray <: DenseArray` (pointer +
    extents + finaliser, copies through the library, Adapt rules) and every descriptor takes its memory kind from the arrays
    it points at (`mem(a)`), never from a literal: a literal 0 next to a device pointer would make the library stage garbage
    from a device address as if it were host memory."""
    mod = _module()
    body = re.search(r"^mutable struct HIPArray\{T, N\} <: DenseArray\{T, N\}\n(.*?)^    function HIPArray", JL, flags=re.S | re.M).group(1)
    fields = dict((f.strip(), t.strip()) for f, t in (ln.split("::") for ln in body.strip().split("\n")))
    assert fields == {"ptr": "Ptr{T}", "dims": "NTuple{N, Int}", "device": "Int32"}, fields
    assert re.search(r"mutable struct HIPArray\{T, N\} <: DenseArray\{T, N\}", JL)
    assert "finalizer(free_slow, a)" in JL
    have = {m.name for m in mod.methods}
    for need in ("Base.size", "Base.pointer", "Base.unsafe_convert", "Base.similar", "Base.copyto!", "Base.Array", "Base.fill!",
                 "Base.getindex", "Base.setindex!", "Adapt.adapt_storage", "mem", "host_lookup_slow"):
        assert need in have, need
    # both directions of Adapt, and both directions + device-to-device of copyto!
    assert re.search(r"Adapt\.adapt_storage\(::Type\{<:HIPArray\}, x::Array\) = HIPArray\(x\)", JL)
    assert re.search(r"Adapt\.adapt_storage\(::Type\{<:Array\}, x::HIPArray\) = Array\(x\)", JL)
    kinds = set(re.findall(r"Int32\((\d)\)\)\s+# RRTMGP_COPY_(\w+)", JL))
    assert kinds == {("1", "H2D"), ("2", "D2H"), ("3", "D2D")}, kinds
    header = open(os.path.join(ROOT, "include", "rrtmgp_hip.h")).read()
    assert re.search(r"RRTMGP_COPY_H2D = 1, RRTMGP_COPY_D2H = 2, RRTMGP_COPY_D2D = 3", header)
    # the memory-kind trait: host for plain arrays, device for HIPArray, followed through views and wrappers
    assert re.search(r"mem\(::AbstractArray\) = Int32\(0\)", JL) and re.search(r"mem\(::HIPArray\) = Int32\(1\)", JL)
    assert re.search(r"mem\(a::Union\{SubArray, PermutedDimsArray, Base\.ReshapedArray\}\) = mem\(parent\(a\)\)", JL)
    # every descriptor constructor that carries a `mem` (or `metric_mem` / `z_mem`) field fills it with mem(...)
    toks = [t for t in JLITE.tokenize(JL) if t.kind != "nl"]
    first_arg = {}
    for i, t in enumerate(toks):
        if t.kind == "id" and t.text in ("AtmosStateDesc", "LwBcsDesc", "SwBcsDesc", "FluxOutDesc", "GrayStateDesc") \
                and toks[i + 1].text == "(" and toks[i - 1].text not in ("struct", "{", "Ref"):
            close = JLITE._matching(toks, i + 1)
            args = JLITE._split_top(toks[i + 2:close], ",")
            first_arg.setdefault(t.text, []).append("".join(x.text for x in args[0]))
    assert first_arg, first_arg
    for name, firsts in first_arg.items():
        assert all(f.startswith("mem(") for f in firsts), (name, firsts)
    assert re.search(r"SolveOpts\(lw_angles\(s\.lws\), mem\(s\.deep_atmosphere_inverse_scaling\)", JL)
    assert re.search(r"isothermal_boundary_layer,\s+mem\(center_z\), lookup_lw\.idx_h2o", JL)
    # the three view-taking methods pass the memory kind of their output array
    for sym, out in (("rrtmgp_hip_compute_col_gas", "col_dry"), ("rrtmgp_hip_compute_relative_humidity", "rh"),
                     ("rrtmgp_hip_compute_gray_heating_rate", "hr_lay")):
        i = JL.index(sym)
        assert re.search(r"workspace\(dev, ncol, nlay, FT, true\), mem\(%s\), ncol, nlay" % out, JL[i:i + 700]), sym
    # lookups built with the device array type come down once before the library re-lays them out
    assert JL.count("lkp = host_lookup_slow(dlkp)") == 3
    # a resident device is one device
    assert re.search(r"resident && length\(ids\) != 1", JL)



def test_resident_arrays_are_allocated_on_the_resident_device_not_on_gpu_0():
    """`ClimaComms.array_type` returns the bare type, so `DA{FT}(undef, ...)` cannot carry the ordinal of
    `HIPDevice(id; resident = true)`; every HIPArray constructor defaulted to `device = 0`.  Now the constructors default to
    the module's resident device, which `HIPDevice(id; resident = true)` sets, and `update_fluxes!` refuses resident arrays that
    do not live on its workspace's GPU."""
    assert "device::Integer = 0" not in JL
    ctors = re.findall(r"^(?:    function )?HIPArray(?:\{[^}]*\})?\([^\n]*device::Integer = ([^\)]+)\)", JL, flags=re.M)
    assert len(ctors) >= 7 and set(ctors) == {"RESIDENT_DEVICE[]"}, ctors
    assert re.search(r"const RESIDENT_DEVICE = Ref\{Int32\}\(0\)", JL)
    new_dev = re.search(r"^function new_device_slow\(.*?^end", JL, flags=re.S | re.M).group(0)
    assert "resident && (RESIDENT_DEVICE[] = Int32(first(ids)))" in new_dev
    step = re.search(r"^function update_fluxes!\(s::HIPSpectralSolver.*?^end", JL, flags=re.S | re.M).group(0)
    assert "dev.resident && check_resident_device(dev, as.layerdata" in step
    chk = re.search(r"^function check_resident_device\(.*?^end", JL, flags=re.S | re.M).group(0)
    assert "dev.ids[1]" in chk and "wrong_device_slow(have, want)" in chk
    # similar / copy keep the device of the array they come from
    assert "Base.similar(a::HIPArray, ::Type{T}, dims::Dims{N}) where {T, N} = HIPArray{T, N}(undef, dims; device = a.device)" in JL


def test_gray_step_override_is_one_library_call_and_is_field_checked():
    """Round 6: `update_fluxes!(s::HIPGraySolver)` = ONE ccall of rrtmgp_hip_update_fluxes_gray with the presentation arrays,
    so gray radiation runs on device-resident HIPArrays (the generic method's presentation copies / net sum are broadcasts).
    The solver alias is followed by the field checker like the spectral one."""
    m = re.search(r"^function update_fluxes!\(s::HIPGraySolver, seedval = nothing\)\n(.*?)^end", JL, flags=re.S | re.M)
    assert m, "no gray override"
    body = m.group(1)
    assert body.count("ccall(") == 1 and ":rrtmgp_hip_update_fluxes_gray" in body
    assert "presented_desc(s.presented_flux_lw, nothing, nothing)" in body and "ptr(s.net_flux_buffer)" in body
    assert "GC.@preserve s sc" in body and "check_resident_device(dev, gs.p_lay" in body
    assert "const HIPGraySolver = RRTMGP.RRTMGPSolver{<:HIPGrid, <:RRTMGP.GrayRadiation}" in JL
    broken = JL.replace("set!(sc.flux_sw, presented_desc(s.presented_flux_sw, nothing, nothing))",
                        "set!(sc.flux_sw, presented_desc(s.presented_sw, nothing, nothing))", 1)
    assert broken != JL and any(why == "RRTMGPSolver has no field presented_sw" for _, _, why in _bad_fields(broken)), _bad_fields(broken)
    broken = JL.replace("gray_desc(gs, RP.Stefan(s.params))", "gray_desc(gs, RP.Stefan(s.param_set))", 1)
    assert broken != JL and any(why == "RRTMGPSolver has no field param_set" for _, _, why in _bad_fields(broken))
