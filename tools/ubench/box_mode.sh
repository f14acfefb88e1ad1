#!/bin/bash
# GPU box: is this one of the boxes where NO set of arrays streams well together (tools/ubench/placement_triples)?  Then which variant
# of the single-step kernel is the fastest here (tools/tune_leapfrog.py)?  Output appended to gpurun_out/box_mode.log
O=gpurun_out/box_mode.log
tools/ubench/placement_triples 12 > gpurun_out/_triples.log 2>&1
echo "=== $(date +%s) ===" >> $O
grep -E "one array alone|good" gpurun_out/_triples.log >> $O
python tools/tune_leapfrog.py 2>&1 | grep "^diag" >> $O
tail -16 $O
