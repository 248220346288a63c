#!/bin/bash
# Builds build_variants/libsfm_gpu_<tag>.so: the product library with ransac_kernels.hip compiled with extra -D flags.
#   scripts/experiments/variant_lib.sh w4 -DSFM_SCORE_WAVES=4
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I $R/include -I $R/sfm_amd/csrc "$@" \
    -c $R/sfm_amd/csrc/ransac_kernels.hip -o $R/build_variants/ransac_kernels_$TAG.o
OBJS=$(ls $R/sfm_amd/csrc/*.o | grep -v timing | grep -v ransac_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/build_variants/libsfm_gpu_$TAG.so $OBJS $R/build_variants/ransac_kernels_$TAG.o
echo $R/build_variants/libsfm_gpu_$TAG.so
