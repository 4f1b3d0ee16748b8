#!/bin/bash
# round 6: the GPU suite again (after the fuzz forms were brought in line with the pruned options), the long differential fuzz with
# the single-launch sampling window, and round 2's latency sweep of the streaming kernels for profiles/r06_small_n.md section 6
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_suite.log 2>&1; echo "rc $?" >> gpurun_out/r06_suite.log
tail -4 gpurun_out/r06_suite.log | cut -c1-200
timeout 900 python tools/fuzz_campaign.py 40 61 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06_fuzz.txt; cat gpurun_out/r06_fuzz.txt
timeout 300 bash tools/latency_sweep.sh run 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_latency_sweep.txt; grep -i "262\|N \|kernel" gpurun_out/r06_latency_sweep.txt | head -40
