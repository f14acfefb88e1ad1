#!/bin/bash
# Diagnostic build with in-kernel cycle stamps -> inplacedhmc.jl_amd/libidhmc_NAME.so; usage: stamps.sh "EXTRA FLAGS" NAME (never shipped/used by tests)
exec "$(dirname "$0")/build_variant.sh" "${2:-stamps}" "-DIDHMC_STAMPS $1"
