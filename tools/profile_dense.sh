#!/bin/bash
# GPU box: rocprofv3 kernel-trace of the configs[3] measurement (tools/bench_dense.py)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dense -- python3 $R/tools/bench_dense.py > $OUT/prof_dense.log 2>&1
find $OUT/prof_dense -name "*kernel_stats.csv"
