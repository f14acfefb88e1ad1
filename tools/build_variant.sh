#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." -> inplacedhmc.jl_amd/libidhmc_NAME.so (experiments only; never shipped)
set -e
cd "$(dirname "$0")/../inplacedhmc.jl_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $2"
D=/tmp/idhmc_var_$1; mkdir -p $D
SRCS=$(sed -n 's/^SRCS = //p' Makefile)
for f in $SRCS; do /opt/rocm/bin/hipcc $F -c $f -o $D/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libidhmc_$1.so $D/*.o -lhiprtc -ldl
