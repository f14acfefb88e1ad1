#!/bin/bash
# GPU box: NUTS transitions as single launches against several per launch (idhmc_nuts_transitions), separable densities, a few shapes.
# Output: gpurun_out/r03_transitions_per_launch_sweep.log
O=gpurun_out/r03_transitions_per_launch_sweep.log
echo "# tools/sweep_fused.sh: tools/bench_nuts.py FUSED=1 (10 single launches; then 2 warm + 10, 10, 40, 160 transitions per launch), shared metric unless noted" > $O
for cfg in "D=1024 C=65536 EPS=0.25" "D=1024 C=65536 EPS=0.03" "D=1024 C=65536 EPS=0.25 METRIC=perchain" "D=1024 C=8192 EPS=0.25" "D=512 C=65536 EPS=0.25" "D=256 C=65536 EPS=0.25" "D=256 C=16384 EPS=0.25" "D=128 C=65536 EPS=0.3"; do
    echo "## $cfg" >> $O
    env $cfg FUSED=1 timeout -k 10 200 python tools/bench_nuts.py 2>&1 | grep "steps/s" | cut -c1-130 >> $O
done
cat $O
