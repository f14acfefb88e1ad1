#!/bin/bash
# usage: dense_streams.sh lib-suffix "K GRID" ...   (GRID 0 = the library's own choice)
cd $GRAFT_REPO_ROOT
export IDHMC_LIB=$PWD/inplacedhmc.jl_amd/libidhmc$1.so; shift
for cfg in "$@"; do
  set -- $cfg
  if [ "$2" = 0 ]; then unset IDHMC_DENSE_GRID; else export IDHMC_DENSE_GRID=$2; fi
  echo "$(basename $IDHMC_LIB) $(K=$1 python3 tools/ubench/dense_streams.py 2>&1 | tail -1)"
done
