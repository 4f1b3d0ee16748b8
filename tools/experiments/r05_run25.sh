#!/bin/bash
# kernel durations of the sampling call (rocprofv3 kernel trace) back to back vs with the GPU idle between calls
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r05k
for gap in 0 100 1000 10000; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05k/kt$gap -o kt -- python $GRAFT_REPO_ROOT/tools/traverse_replay.py $GRAFT_REPO_ROOT/profiles/r02_sampling_state.npz 60 --gap-us=$gap > /dev/null 2>&1)
  echo "gap $gap us:"; python tools/kernel_summary.py gpurun_out/r05k/kt$gap | grep "traverse_" | cut -c1-40,100-150
  rm -rf gpurun_out/r05k/kt$gap
done
