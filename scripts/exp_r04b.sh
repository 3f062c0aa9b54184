#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_grid.py tests/test_estimator_golden.py -m gpu -q -x --timeout 600 > gpurun_out/r04b_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r04b_tests.log
tail -15 gpurun_out/r04b_tests.log
(python scripts/time_tail_select.py; python scripts/time_tail_select.py) > gpurun_out/r04b_tail.log 2>&1
tail -3 gpurun_out/r04b_tail.log
