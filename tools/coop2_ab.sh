#!/bin/bash
# dense NUTS at configs[3]: one chain per wavefront (k_nuts<DenseMvnCoop>) vs two (k_nuts_coop2), same process conditions, twice each
mkdir -p gpurun_out
for rep in 1 2; do
  for v in 0 1; do
    echo "IDHMC_DENSE_COOP2=$v rep $rep: $(IDHMC_DENSE_COOP2=$v NUTS_ONLY=1 python tools/bench_dense.py 2>&1 | grep 'dense NUTS')"
  done
done
for c in 8192 32768 65536; do
  for v in 0 1; do
    echo "C=$c IDHMC_DENSE_COOP2=$v: $(C=$c IDHMC_DENSE_COOP2=$v NUTS_ONLY=1 python tools/bench_dense.py 2>&1 | grep 'dense NUTS')"
  done
done
