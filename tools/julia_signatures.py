#!/usr/bin/env python3
"""Extracts, from the reference's CUDA extension, the positional signature of every device method an
extension must add, and the flux-layout rule of `_coalesced_2d`, as DATA:

    python tools/julia_signatures.py /root/reference > tests/golden/julia_signatures.json

The JSON holds method names, parameter names, type annotations and default flags only (no source
text).  tests/test_julia_binding.py compares ext/RRTMGPHIPExt.jl against it."""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import julia_lite as JL  # noqa: E402


_ID = r"[^\W\d][\w!]*"  # Julia identifiers may carry Greek letters and subscripts (\w is Unicode-aware)


def extract_structs(ref):
    """{struct name: {"file", "line", "params": {type parameter: bound or ""}, "fields": [[name, declared type]]}} for
    every struct under src/: names and annotations only."""
    structs = {}
    for path in sorted(glob.glob(os.path.join(ref, "src", "**", "*.jl"), recursive=True)):
        lines = open(path).read().split("\n")
        i = 0
        while i < len(lines):
            m = re.match(r"^(?:mutable\s+)?struct\s+(" + _ID + r")", lines[i])
            if not m:
                i += 1
                continue
            name, line0, hdr = m.group(1), i + 1, lines[i]
            if re.search(r"\bend\s*$", hdr):  # one-line singleton
                structs[name] = {"file": os.path.relpath(path, ref), "line": line0, "params": {}, "fields": []}
                i += 1
                continue
            while hdr.count("{") > hdr.count("}") or hdr.rstrip().endswith("<:"):
                i += 1
                hdr += " " + lines[i].strip()
            params = {}
            b0 = hdr.find("{")
            if b0 >= 0 and (hdr.find("<:") < 0 or b0 < hdr.find("<:")):
                depth, b1 = 0, b0
                for b1 in range(b0, len(hdr)):  # the brace that closes the parameter list
                    depth += (hdr[b1] == "{") - (hdr[b1] == "}")
                    if depth == 0:
                        break
                depth, cur, parts = 0, "", []
                for c in hdr[b0 + 1:b1]:
                    depth += (c in "{(") - (c in "})")
                    if c == "," and depth == 0:
                        parts.append(cur); cur = ""
                    else:
                        cur += c
                parts.append(cur)
                for part in parts:
                    part = part.strip()
                    if part:
                        nm, _, bound = part.partition("<:")
                        params[nm.strip()] = bound.strip()
            fields, depth = [], 0
            i += 1
            while i < len(lines) and not (lines[i].startswith("end") and depth == 0):
                t = lines[i].strip()
                if re.match(r"^(function|for|if|while|let|begin)\b", t):
                    depth += 1
                elif t == "end" and depth > 0:
                    depth -= 1
                elif depth == 0 and not t.startswith(('"', "#")):
                    fm = re.match(r"^(" + _ID + r")\s*(?:::\s*(.+?))?\s*(?:#.*)?$", t)
                    if fm:
                        fields.append([fm.group(1), fm.group(2) or ""])
                i += 1
            structs[name] = {"file": os.path.relpath(path, ref), "line": line0, "params": params, "fields": fields}
            i += 1
    return structs


def main(ref):
    out = {"source": "CliMA/RRTMGP.jl ext/cuda/*.jl, src/optics/Fluxes.jl (method headers only); src/**/*.jl (struct field names)",
           "methods": [], "coalesced_2d": [], "structs": extract_structs(ref)}
    for path in sorted(glob.glob(os.path.join(ref, "ext", "cuda", "*.jl"))):
        mod = JL.parse_module(open(path).read())
        for m in mod.methods:
            if not m.params or "CUDADevice" not in m.params[0].type:
                continue
            out["methods"].append({
                "file": os.path.relpath(path, ref), "line": m.line, "name": m.name,
                "where": m.where,
                "params": [{"name": p.name, "type": JL.norm_type(p.type), "default": p.has_default, "vararg": p.vararg}
                           for p in m.params]})
    # _coalesced_2d(DA, FT, d1, d2): which array type gets which physical layout
    mod = JL.parse_module(open(os.path.join(ref, "src", "optics", "Fluxes.jl")).read())
    for m in mod.methods:
        if m.name != "_coalesced_2d":
            continue
        body = "".join(t.text for t in m.body if t.kind != "nl")
        dims = re.search(r"\(undef,(d\d),(d\d)\)", body)
        out["coalesced_2d"].append({
            "line": m.line, "array_type": JL.norm_type(m.params[0].type) or "any",
            "wrapper": "PermutedDimsArray" if "PermutedDimsArray" in body else "none",
            # kernels index (d1, d2) = (ncol, nlev); the parent is allocated as:
            "parent_dims": [dims.group(1), dims.group(2)] if dims else None})
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
