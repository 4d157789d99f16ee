#!/bin/bash
# Build libscone_hip.so for gfx950 in-tree (scone_gcn_amd/libscone_hip.so).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libscone_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-result ${SCN_EXTRA_FLAGS:-}"
mkdir -p "$HERE/build"
pids=()
for f in scn_conv scn_blocked scn_readout scn_dense scn_small; do
  src="$HERE/$f.hip"; obj="$HERE/build/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/scn_internal.h" -nt "$obj" ] || [ -n "$(find "$HERE" -name "*.inc" -newer "$obj")" ] || [ "$ROOT/include/scone_hip.h" -nt "$obj" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/build/scn_conv.o" "$HERE/build/scn_blocked.o" "$HERE/build/scn_readout.o" "$HERE/build/scn_dense.o" "$HERE/build/scn_small.o"
echo "built $OUT"
