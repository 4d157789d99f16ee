#!/bin/bash
# usage: tools/ab_build.sh <name> "<extra hipcc flags>" : diagnostic build of the library with extra -D flags into
# tools/ab/lib_<name>.so (git-ignored; select it with SCN_LIB_PATH) -- same-box A/B runs of kernel variants.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
C="$ROOT/scone_gcn_amd/csrc"
NAME=$1; EXTRA=${2:-}
mkdir -p "$ROOT/tools/ab/obj_$NAME"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$C -Wall -Wno-unused-result $EXTRA"
for f in scn_conv scn_blocked scn_readout scn_dense scn_small; do
  /opt/rocm/bin/hipcc $FLAGS -c "$C/$f.hip" -o "$ROOT/tools/ab/obj_$NAME/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/lib_$NAME.so" "$ROOT"/tools/ab/obj_$NAME/*.o
rm -rf "$ROOT/tools/ab/obj_$NAME"
echo "built tools/ab/lib_$NAME.so"
