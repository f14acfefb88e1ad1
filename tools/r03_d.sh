python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sweep.py tests/test_gpu_warmup.py tests/test_gpu_edges.py tests/test_gpu_custom.py -x -q -m gpu 2>&1 | tail -3
EPS=0.25 NT=10 python tools/bench_nuts.py 2>&1 | grep steps/s
EPS=0.03 NT=5 python tools/bench_nuts.py 2>&1 | grep steps/s
METRIC=perchain EPS=0.25 NT=10 python tools/bench_nuts.py 2>&1 | grep steps/s
METRIC=perchain EPS=0.03 NT=5 python tools/bench_nuts.py 2>&1 | grep steps/s
