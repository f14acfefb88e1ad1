#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/bench_dense.py (NUTS part): one k_nuts<DenseMvnCoop> launch per transition against
# 5 / 20 / 50 transitions per launch (idhmc_nuts_transitions).  Output: gpurun_out/prof_dense_nuts_*; summary printed.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
export NUTS_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dense_nuts -- python3 $R/tools/bench_dense.py > $OUT/prof_dense_nuts.log 2>&1
cat $OUT/prof_dense_nuts.log | grep "dense NUTS"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_dense_nuts/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_nuts" in r["Kernel_Name"]]
print("k_nuts launches:", len(rows))
for r in rows:
    print("  %.3f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
