set -x
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1 && tail -5 gpurun_out/r03_profile_bench.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['frac'], d['state_placement']); print(json.dumps(d.get('cfg3_full'), indent=1)[:3000])"
rm -rf gpurun_out/prof_r03_trace gpurun_out/prof_r03_fetch gpurun_out/prof_r03_write
