#!/bin/bash
# GPU box: kernel-trace of the dense path (configs[3]); summary lands in gpurun_out/prof_dense
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dense -- python3 $R/tools/bench_dense.py > $OUT/prof_dense.log 2>&1
cat $OUT/prof_dense.log
