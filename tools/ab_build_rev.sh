#!/bin/bash
# usage: tools/ab_build_rev.sh <name> <git rev> ["<extra hipcc flags>"] : the library as it was at <rev> (sources from git, nothing in the
# working tree is touched) into tools/ab/lib_<name>.so -- the "before" side of a same-box A/B against the working tree (tools/ab_run.py).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; REV=$2; EXTRA=${3:-}
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" scone_gcn_amd/csrc include | tar -x -C "$TMP"
C="$TMP/scone_gcn_amd/csrc"
mkdir -p "$ROOT/tools/ab" "$TMP/obj"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$TMP/include -I$C -Wall -Wno-unused-result $EXTRA"
for f in scn_conv scn_blocked scn_readout scn_dense scn_small; do
  /opt/rocm/bin/hipcc $FLAGS -c "$C/$f.hip" -o "$TMP/obj/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/lib_$NAME.so" "$TMP"/obj/*.o
echo "built tools/ab/lib_$NAME.so from $REV"
