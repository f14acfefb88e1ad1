#!/bin/bash
# GPU box: rocprofv3 kernel-trace of the cfg3-style NUTS measurement (tools/bench_nuts.py)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_nuts -- python3 $R/tools/bench_nuts.py > $OUT/prof_nuts.log 2>&1
find $OUT/prof_nuts -name "*kernel_stats.csv"
