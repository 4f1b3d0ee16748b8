#!/bin/bash
# build a variant of libnerfacc_hip.so with extra -D flags into tools/_prof/ (git-ignored; travels with gpurun):
#   tools/build_variant.sh <name> [-DFLAG ...]   ->  tools/_prof/libnerfacc_hip_<name>.so
# use with  NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=tools/_prof/libnerfacc_hip_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
mkdir -p "$ROOT/tools/_prof"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I"$ROOT/include" "$@" -shared \
    "$ROOT"/nerfacc_amd/csrc/*.hip -o "$ROOT/tools/_prof/libnerfacc_hip_$name.so"
echo "$ROOT/tools/_prof/libnerfacc_hip_$name.so"
