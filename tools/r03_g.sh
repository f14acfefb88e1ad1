python -m pytest tests/ -m gpu -x -q 2>&1 | tail -3
for e in 0.25 0.03; do echo -n "shared "; EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; echo -n "perchain "; METRIC=perchain EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done
python bench.py --no-cpu --no-cfg3 --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['nuts']['leapfrog_steps_per_s'], d['nuts']['deep_trees']['leapfrog_steps_per_s'], d['dense']['nuts']['leapfrog_steps_per_s'], d['dense']['single_step_sweeps']['chain_steps_per_s'])"
