#!/bin/bash
# Both shipped libraries (what __graft_entry__.build() makes): run before every gpurun — the .so files travel with the snapshot.
set -e
cd "$(dirname "$0")/.."
make -C rrtmgp.jl_amd/csrc -j4 2>&1 | grep -E "error|warning: unused|Error" || true
make -C rrtmgp.jl_amd/csrc -j4 fast 2>&1 | grep -E "error|Error" || true
make -C oracle 2>&1 | grep -E "error" || true
ls -la rrtmgp.jl_amd/*.so
