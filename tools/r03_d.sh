set -o pipefail
python -m pytest tests/ -m gpu -x -q 2>&1 | tail -3 || exit 1
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -5 gpurun_out/r03_bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['frac'], d['nuts']['leapfrog_steps_per_s'], d['nuts']['deep_trees']['leapfrog_steps_per_s']); f=d['cfg3_full']; print(f['warmup']['seconds'], f['warmup']['leapfrog_steps_per_s'], f['sampling']['seconds'], f['sampling']['leapfrog_steps_per_s'], f['chains_bit_identical']); print(d['dense']['single_step_sweeps']['chain_steps_per_s'], d['dense']['steps_fused_64_per_call']['chain_steps_per_s'], d['dense']['nuts']['leapfrog_steps_per_s'], d['dense']['single_step_sweeps']['lanes'])"
