#!/bin/bash
# tools/build_coop2_variant.sh NAME "-DFLAG ..." -> inplacedhmc.jl_amd/libidhmc_NAME.so with only idhmc_nuts_coop2.hip rebuilt
# (the other objects are the shipped build's; experiments only, never shipped)
set -e
cd "$(dirname "$0")/../inplacedhmc.jl_amd/csrc"
make -s -j8 >/dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $2"
/opt/rocm/bin/hipcc $F -c idhmc_nuts_coop2.hip -o /tmp/coop2_$1.o
OBJS=$(sed -n 's/^SRCS = //p' Makefile | sed 's/\.hip/.o/g; s/idhmc_nuts_coop2\.o//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libidhmc_$1.so $OBJS /tmp/coop2_$1.o -lhiprtc -ldl
