#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 --durations=15 > gpurun_out/r04c_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r04c_tests.log
tail -30 gpurun_out/r04c_tests.log
