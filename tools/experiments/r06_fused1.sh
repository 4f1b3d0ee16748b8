#!/bin/bash
# round 6: the fused sampling launch — tests, then the replay of the bench's steady state fused / unfused, with a kernel trace
export TMPDIR=/tmp
O=gpurun_out/r06_fused1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_sampling.py -x -q 2>&1 | tail -15 > $O/tests.log
for f in 1 0; do
  NFA_FUSED_SAMPLE=$f python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 --check > $O/replay_f$f.log 2>&1
  NFA_FUSED_SAMPLE=$f python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 --gap-us=200 > $O/replay_gap_f$f.log 2>&1
  D=$(mktemp -d /tmp/ktXXXX)
  NFA_FUSED_SAMPLE=$f rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 > /dev/null 2>&1
  python tools/kernel_summary.py $D > $O/kstats_f$f.txt 2>&1
done
cat $O/tests.log $O/replay_f1.log $O/replay_f0.log $O/replay_gap_f1.log $O/replay_gap_f0.log; grep nfa $O/kstats_f1.txt $O/kstats_f0.txt
