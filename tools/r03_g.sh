python -m pytest tests/ -m gpu -x -q 2>&1 | tail -2
python tools/bench_accum.py 2>&1 | tail -6
