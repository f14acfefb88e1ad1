#!/bin/bash
# GPU box: kernel-trace + PMC passes of the bench command; summaries land in gpurun_out/prof_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 200 --warmup 20 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- $CMD > $OUT/prof_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- $CMD > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- $CMD > $OUT/prof_write.log 2>&1
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20
