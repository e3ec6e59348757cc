#!/usr/bin/env python3
"""Loop census of a column kernel's gfx950 assembly: for every loop of at least --min instructions, the instruction mix
(VALU, transcendental, DPP, scratch = spill traffic, global loads / stores, LDS, s_waitcnt, s_nop).

    python tools/isa_loops.py solve_lw 'lw_noscat_kernel<float, 3, 0>' [--min 60] [-- extra hipcc flags]

Compiles rrtmgp.jl_amd/csrc/<file>.hip with the Makefile's flags to assembly (device only) and prints one line per loop
of the kernels whose demangled name contains the pattern."""
import os
import re
import subprocess
import sys

ROOT = os.environ.get("ISA_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize"]


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    mn = 60
    if "--min" in args:
        i = args.index("--min"); mn = int(args[i + 1]); del args[i:i + 2]
    src, pat = args[0], args[1]
    out = f"/tmp/isa_{src}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", out, f"{src}.hip"],
                   cwd=os.path.join(ROOT, "rrtmgp.jl_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"(_ZN6rrtmgp\w+):", l)] if m]
    for n, (i0, mangled) in enumerate(starts):
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"rrtmgp::|\(rrtmgp::\w+<\w+>\)|void ", "", name)
        if pat not in name:
            continue
        i1 = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
        body = lines[i0:i1]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        whole = [x.strip() for x in body if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        print(name, f"| {len(whole)} instr, static v_readlane {sum(x.startswith('v_readlane') for x in whole)} v_writelane {sum(x.startswith('v_writelane') for x in whole)} "
                    f"flat {sum(x.startswith('flat_') for x in whole)} s_load {sum(x.startswith('s_load') for x in whole)} scratch {sum(x.startswith('scratch_') for x in whole)}")
        for a, b in sorted(set(loops)):
            ins = [x.strip() for x in body[a:b + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            if len(ins) < mn:
                continue
            c = lambda p: sum(1 for x in ins if re.match(p, x))  # noqa: E731
            print(f"  loop lines {a}-{b}: {len(ins)} instr | VALU {c(r'v_')} (trans {c(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)')}, dpp {sum('dpp' in x for x in ins)}, "
                  f"f64 {c(r'v_[a-z0-9_]+_f64')}) SALU {c(r's_(?!waitcnt|nop|cbranch|branch|barrier)')} | scratch {c(r'scratch_')} "
                  f"gload {c(r'global_load')} gstore {c(r'global_store')} flat {c(r'flat_')} sload {c(r's_load')} ds {c(r'ds_')} | waitcnt {c(r's_waitcnt')} nop {c(r's_nop')} "
                  f"barrier {c(r's_barrier')} | readlane {c(r'v_readlane')} writelane {c(r'v_writelane')}")


if __name__ == "__main__":
    main()
