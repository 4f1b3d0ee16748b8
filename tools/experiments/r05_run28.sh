#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 40 --check 2>&1 | grep "oracle\|^rays" | cut -c1-110; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05l -o kt -- python $GRAFT_REPO_ROOT/tools/traverse_replay.py $GRAFT_REPO_ROOT/profiles/r02_sampling_state.npz 60 > /dev/null 2>&1)
python tools/kernel_summary.py gpurun_out/r05l | grep "traverse_" | cut -c1-40,100-150; rm -rf gpurun_out/r05l
timeout 300 python -m pytest tests/test_k2_reference.py -x -q -m gpu -p no:cacheprovider -k "m1_sphere or lego_4k or lego_12k" 2>&1 | tail -1
