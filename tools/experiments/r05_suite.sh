#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_suite.log 2>&1; echo "rc $?" >> gpurun_out/r05_suite.log
tail -6 gpurun_out/r05_suite.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
