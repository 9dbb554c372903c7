#!/bin/bash
# Builds a VARIANT of libls1hip.so for same-box A/B timing runs: tools/ab_variant.sh <tag> "<extra hipcc flags>" [file.hip ...]
# The named sources (default: csrc/kernels_force_verlet.hip) are recompiled with the extra flags, everything else is taken from
# the regular build; output: ls1-mardyn_amd/lib/variants/libls1hip_<tag>.so (git-ignored, travels with gpurun).
# Every object compiled here carries -DLS1_BUILD_VARIANT: it may pull timing hooks with wrong physics (csrc/variants/) and marks the
# library (option "build_variant" = 1, "+variant" in ls1hip_version); the regular Makefile never sets it.
# Use: LS1HIP_LIB=ls1-mardyn_amd/lib/variants/libls1hip_<tag>.so python bench.py ...
set -e
tag=$1; flags=$2; shift 2 || true
files=${@:-csrc/kernels_force_verlet.hip}
cd "$(dirname "$0")/../ls1-mardyn_amd"
make -s -j4
mkdir -p build_$tag lib/variants
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  if echo " $files " | grep -q "csrc/$b.hip"; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value -DLS1_BUILD_VARIANT=1 $flags -c csrc/$b.hip -o build_$tag/$b.o
    objs="$objs build_$tag/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o lib/variants/libls1hip_$tag.so
echo "built lib/variants/libls1hip_$tag.so"
