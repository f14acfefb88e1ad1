# round-3 evidence in one call: headline profile passes, k_nuts counters, default bench line (GPU box)
set -x
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1 && tail -2 gpurun_out/r03_profile_bench.log
rm -rf gpurun_out/prof_r03_trace gpurun_out/prof_r03_fetch gpurun_out/prof_r03_write
bash tools/profile_nuts_pmc.sh > gpurun_out/r03_nuts_pmc.log 2>&1; tail -2 gpurun_out/r03_nuts_pmc.log
python3 tools/summarize_nuts_pmc.py gpurun_out/nuts_pmc gpurun_out/r03_nuts_pmc.json > /dev/null; rm -rf gpurun_out/nuts_pmc/*/*/
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python __graft_entry__.py smoke 2>&1 | tail -1
