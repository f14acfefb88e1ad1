set -o pipefail
python -m pytest tests/ -m gpu -x -q 2>&1 | tail -5 || exit 1
python bench.py --no-cpu > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err || { tail -5 gpurun_out/r03_d_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_d_bench.json')); print(d['value'], d['state_placement']['kind'], d['nuts']['leapfrog_steps_per_s'], d['nuts']['deep_trees']['leapfrog_steps_per_s']); f=d['cfg3_full']; print(f['warmup']['seconds'], f['warmup']['leapfrog_steps_per_s'], f['sampling']['leapfrog_steps_per_s'], f.get('large_run_equals_small_run_bitwise'), f['acceptance_mean'], f['rhat_max'], f['ess_per_draw_min']); print(d['dense']['single_step_sweeps']['kernel_ms'], d['dense']['nuts']['leapfrog_steps_per_s'], d['global_eps_warmup']['allreduces'])"
