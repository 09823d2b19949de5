#!/bin/bash
# tools/build_variant.sh NAME [-D...]: an A/B build of the HIP library with extra compiler flags into
# build/variants/NAME.so (untracked; travels to the GPU box with the snapshot).  Select it with
# DES_HIP_LIB=build/variants/NAME.so (dynearthsol_amd.load_hip_lib).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/dynearthsol_amd/csrc"
mkdir -p "$root/build/variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-pass-failed -I../../include "$@" \
      -c -o "$root/build/variants/$name.o" des_dev.hip
[ -f des_dev2d.o ] || make -s des_dev2d.o
hipcc --offload-arch=gfx950 -shared -o "$root/build/variants/$name.so" "$root/build/variants/$name.o" des_dev2d.o -L/opt/rocm/lib -lrccl
rm -f "$root/build/variants/$name.o"
echo "built build/variants/$name.so"
