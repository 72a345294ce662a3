#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05y
( time timeout 2400 python -m pytest tests/ -m gpu -q -p no:cacheprovider > gpurun_out/r05y/suite.log 2>&1 ) 2> gpurun_out/r05y/suite.time; echo "suite rc $?"; tail -5 gpurun_out/r05y/suite.log; tail -3 gpurun_out/r05y/suite.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05y/smoke.log 2>&1; tail -3 gpurun_out/r05y/smoke.log
