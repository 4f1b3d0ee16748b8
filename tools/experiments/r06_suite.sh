#!/bin/bash
# the GPU suite, the smoke call and the fallback tests at the current kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_suite.log 2>&1; echo "rc $?" >> gpurun_out/r06_suite.log
tail -6 gpurun_out/r06_suite.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
