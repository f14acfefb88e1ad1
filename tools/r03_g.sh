python -m pytest tests/ -m gpu -x -q 2>&1 | tail -3
for e in 0.25 0.03; do echo -n "shared "; EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; echo -n "perchain "; METRIC=perchain EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done
python tools/bench_dense.py 2>&1 | tail -3
