#!/bin/bash
# usage: nuts_ab.sh lib-suffix ... ; runs bench_nuts at depth 4 and 7 for each variant (WIDE env passes through)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  for e in ${EPSLIST:-0.25 0.03}; do
    L=inplacedhmc.jl_amd/libidhmc$v.so
    echo "== $v wide=${IDHMC_NUTS_WIDE:-auto} eps=$e: $(IDHMC_LIB=$PWD/$L EPS=$e NT=5 python3 tools/bench_nuts.py | grep steps/s | cut -c1-150)"
  done
done
