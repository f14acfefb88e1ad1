#!/bin/bash
# GPU box: kernel-trace of the whole configs[2] run (default warm-up + 200 draws, per-chain eps and metric); summary in gpurun_out/prof_cfg3
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg3 -- python3 $R/tools/run_cfg3.py > $OUT/prof_cfg3.log 2>&1
grep -v "^W2026\|^E2026" $OUT/prof_cfg3.log | tail -15
