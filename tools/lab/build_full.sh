#!/bin/bash
# tools/lab/build_full.sh <tag> [-Dflags...]: the whole product library as a variant in tools/lab/_build/<tag>/libpp_hip.so
# (use with PP_HIP_LIB=tools/lab/_build/<tag>/libpp_hip.so python tools/bench_*.py)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; shift
out=$root/tools/lab/_build/$tag
mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared -ffp-contract=off --offload-arch=gfx950 -I"$root/include" \
  -Wl,-rpath,/opt/rocm/lib "$@" "$root"/3d-object-detection_amd/csrc/pp_*.hip -o "$out/libpp_hip.so"
echo "built $out/libpp_hip.so"
