set -x
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1 && tail -3 gpurun_out/r03_profile_bench.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_r03_trace gpurun_out/prof_r03_fetch gpurun_out/prof_r03_write
python -m pytest tests/ -m gpu -x -q 2>&1 | tail -2
bash tools/profile_bench.sh r03b > gpurun_out/r03b_profile_bench.log 2>&1 && tail -3 gpurun_out/r03b_profile_bench.log
rm -rf gpurun_out/prof_r03b_trace gpurun_out/prof_r03b_fetch gpurun_out/prof_r03b_write
python -c "
import json
for t in ('r03','r03b'):
    d=json.load(open('gpurun_out/%s_leapfrog_pmc.json'%t)); print(t, d['avg_duration_ns'], d['placement_mode'], d['frac_of_8000_from_profile'], d['same_run_bench_line']['frac'], d['hbm_over_algorithmic'], d['other_kernels'].get('k_xchg_sum<0>'))
d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['frac'], d['roofline']['profile_kernel_ms'], d['state_placement']['candidates_tried'], d['cpu_baseline']['value'])
"
