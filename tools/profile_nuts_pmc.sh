#!/bin/bash
# GPU box: HBM traffic of the NUTS kernel (PMC passes, one counter per run as the guide prescribes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
export EPS=${EPS:-0.03} NT=${NT:-3}
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_nuts_fetch -- python3 $R/tools/bench_nuts.py > $OUT/prof_nuts_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_nuts_write -- python3 $R/tools/bench_nuts.py > $OUT/prof_nuts_write.log 2>&1
grep "steps/s" $OUT/prof_nuts_fetch.log | cut -c1-160
