#!/bin/bash
# Full GPU suite with the complete log kept under gpurun_out/ (a failing run must leave its evidence):
#   gpurun -- tools/gpu_tests.sh
mkdir -p gpurun_out
stamp=$(date +%H%M%S)
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests_$stamp.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_tests_$stamp.log
grep -E "^(FAILED|ERROR)" gpurun_out/gpu_tests_$stamp.log
exit $rc
