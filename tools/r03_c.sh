python -m pytest tests/test_gpu_multirank.py tests/test_gpu_warmup.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cfg3 --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03_c_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_c_prof.err
cd $GRAFT_REPO_ROOT && grep -h "k_xchg_sum\|k_spin" gpurun_out/prof_c/*/*_kernel_stats.csv; rm -rf gpurun_out/prof_c
