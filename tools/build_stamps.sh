#!/bin/bash
# Diagnostic build with s_memtime stamps in fwd_c32_kernel / bwd_c32(_bf16)_kernel -> tools/ubench/libscone_hip_stamps.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"; ROOT="$(cd "$HERE/.." && pwd)"; C="$ROOT/scone_gcn_amd/csrc"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$C -DSCN_STAMPS ${SCN_EXTRA_FLAGS:-}"
objs=()
for f in scn_conv scn_blocked scn_readout scn_dense scn_small; do /opt/rocm/bin/hipcc $FLAGS -c "$C/$f.hip" -o "/tmp/stamps_$f.o" & objs+=("/tmp/stamps_$f.o"); done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/ubench/libscone_hip_stamps.so" "${objs[@]}"
echo built
