#!/bin/bash
# gpurun helper: GPU test suite, log to gpurun_out/
mkdir -p gpurun_out
if [ $# -eq 0 ]; then set -- tests; fi
python -m pytest "$@" -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
