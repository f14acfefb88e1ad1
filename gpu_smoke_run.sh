#!/bin/bash
# helper for gpurun: run gpu parity tests
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tee gpurun_out/parity.log | tail -40
