#!/bin/bash
# Run selected GPU tests: bash tools/gpu_test.sh <pytest args>
set -u
mkdir -p gpurun_out
timeout ${SUITE_TIMEOUT:-900} python -m pytest "$@" -m gpu -o faulthandler_timeout=${PER_TEST_TIMEOUT:-300} -q --no-header -rA -s -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit: $?"
grep -E "^(FAILED|ERROR)|passed|failed|hip vs|rel |Error" gpurun_out/pytest_sel.log | tail -30
