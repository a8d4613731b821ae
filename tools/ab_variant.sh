#!/bin/bash
# Build the WORKING TREE's library with extra compiler flags next to the product build, for A/B runs on ONE GPU box:
#   bash tools/ab_variant.sh <name> "<extra flags>"   -> gpurun_tmp_libs/<name>/distaff_amd/libdistaff_hip.so   (git-ignored, travels with gpurun)
#   DISTAFF_HIP_LIB=gpurun_tmp_libs/<name>/distaff_amd/libdistaff_hip.so python bench.py ...
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=/tmp/ab_variant_$name
rm -rf "$root/gpurun_tmp_libs/$name" "$src"
mkdir -p "$root/gpurun_tmp_libs/$name/distaff_amd" "$src"
(cd "$root" && tar -c --exclude='.pytest_cache' distaff_amd/csrc include) | tar -x -C "$src"
make -C "$src/distaff_amd/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-value $flags" > /dev/null 2> "$src/build.err" || { tail -20 "$src/build.err"; exit 1; }
cp "$src/distaff_amd/libdistaff_hip.so" "$root/gpurun_tmp_libs/$name/distaff_amd/"
echo "$flags" > "$root/gpurun_tmp_libs/$name/FLAGS"
echo "variant $name built with: $flags"
