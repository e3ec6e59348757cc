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


def main(ref):
    out = {"source": "CliMA/RRTMGP.jl ext/cuda/*.jl, src/optics/Fluxes.jl (method headers only)", "methods": [], "coalesced_2d": []}
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
