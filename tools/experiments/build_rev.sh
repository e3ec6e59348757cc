#!/bin/bash
# Builds the library of another git revision into rrtmgp.jl_amd/variants/<name>.so (A/B against the working tree in one
# gpurun session).  Usage: tools/experiments/build_rev.sh <git-rev> <name>
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" rrtmgp.jl_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/rrtmgp.jl_amd/csrc" -j6 OUT="$ROOT/rrtmgp.jl_amd/variants/$NAME.so" > /tmp/build_rev_$NAME.log 2>&1 || { tail -20 /tmp/build_rev_$NAME.log; exit 1; }
rm -rf "$TMP"
echo "built variants/$NAME.so from $REV"
