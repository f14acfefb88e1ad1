#!/bin/bash
# GPU box: kernel-trace + the two PMC passes of the bench command (the program directly after `--`); outputs under gpurun_out/prof_<tag>_*.
# usage: tools/profile_bench.sh <tag>      then: python3 tools/summarize_prof.py <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
ARGS="--steps 200 --warmup 20 --no-cpu --no-cfg3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- python3 $R/bench.py $ARGS > $OUT/prof_${TAG}_trace.json 2> $OUT/prof_${TAG}_trace.log
echo "trace pass done" 
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > $OUT/prof_${TAG}_fetch.json 2> $OUT/prof_${TAG}_fetch.log
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $R/bench.py $ARGS > $OUT/prof_${TAG}_write.json 2> $OUT/prof_${TAG}_write.log
echo "write pass done"
cd $R && python3 tools/summarize_prof.py $TAG --stage-only
