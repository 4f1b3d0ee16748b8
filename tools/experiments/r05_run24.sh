#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r05j
timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 40 --check 2>&1 | grep "oracle\|^rays" | cut -c1-150
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05j/kt -o kt -- python $GRAFT_REPO_ROOT/tools/traverse_replay.py $GRAFT_REPO_ROOT/profiles/r02_sampling_state.npz 60 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/kernel_summary.py gpurun_out/r05j/kt | grep "traverse_" | cut -c1-150
rm -rf gpurun_out/r05j/kt
timeout 300 python -m pytest tests/test_gpu_grid.py tests/test_gpu_semantics.py tests/test_gpu_testmode.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
