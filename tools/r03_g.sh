timeout -k 10 300 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 120 python tools/bench_dense.py 2>&1 | tail -3
