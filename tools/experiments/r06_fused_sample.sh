#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_fused_sample; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_sampling.py tests/test_k2_reference.py -x -q 2>&1 | tail -8 > $O/tests.log
timeout 300 python tools/fuse_trace.py profiles/r02_sampling_state.npz 20 > $O/fuse_trace.log 2>&1
for f in 1 0; do
  NFA_FUSED_SAMPLE=$f python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 --check > $O/replay_f$f.log 2>&1
done
D=$(mktemp -d /tmp/ktXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 > /dev/null 2>&1
python tools/kernel_summary.py $D > $O/kstats_f1.txt 2>&1
cat $O/tests.log $O/fuse_trace.log $O/replay_f1.log $O/replay_f0.log; grep nfa $O/kstats_f1.txt | head -4
