#!/bin/bash
# Diagnostic build with in-kernel cycle stamps -> inplacedhmc.jl_amd/libidhmc_NAME.so; usage: stamps.sh "EXTRA FLAGS" NAME (never shipped/used by tests)
set -e
cd "$(dirname "$0")/../inplacedhmc.jl_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DIDHMC_STAMPS $1"
N=${2:-stamps}
mkdir -p /tmp/idhmc_stamps
for f in idhmc_api idhmc_kernels idhmc_nuts idhmc_dense idhmc_dense_mfma idhmc_jit idhmc_comm; do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/idhmc_stamps/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libidhmc_$N.so /tmp/idhmc_stamps/*.o -lhiprtc -ldl
