#!/bin/bash
# tools/lab/build.sh <tag> [-Dflags...]: variant lib (runtime + voxelizer only) and the lab binary in tools/lab/_build/<tag>/
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; shift
out=$root/tools/lab/_build/$tag
mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared -ffp-contract=off --offload-arch=gfx950 -I"$root/include" "$@" \
  "$root/3d-object-detection_amd/csrc/pp_runtime.hip" "$root/3d-object-detection_amd/csrc/pp_voxelize.hip" -o "$out/libpp_lab.so"
/opt/rocm/bin/hipcc -O2 -std=c++17 -I"$root/include" "$root/tools/lab/vox_lab.cpp" -o "$out/vox_lab" -L"$out" -lpp_lab -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo "built $out"
