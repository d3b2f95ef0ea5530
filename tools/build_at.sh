#!/bin/bash
# Build libfcamd.so of another commit next to the working tree's, for tools/ab_lib.py (two LIBRARIES in one process on identical
# buffers -- knobs inside one build do not show what the build itself costs):
#     tools/build_at.sh <commit> [out.so]        default out: tools/_ab/libfcamd_<commit>.so   (git-ignored; travels with gpurun)
#     gpurun -- 'AB_SPARSE=1 python tools/ab_lib.py tools/_ab/libfcamd_<commit>.so fenics-constitutive_amd/lib/libfcamd.so 100000000'
set -eu
C=${1:?commit}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${2:-$R/tools/_ab/libfcamd_$C.so}
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
git -C "$R" archive "$C" fenics-constitutive_amd/csrc include | tar -x -C "$T"
mkdir -p "$(dirname "$OUT")"
cd "$T/fenics-constitutive_amd/csrc"
SRC=$(ls *.hip *.cpp)
# shellcheck disable=SC2086
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -Wall -Wno-unused-function -o "$OUT" $SRC
echo "$OUT"
