#!/bin/bash
# round 6: the single-launch sampling call — tests (fixtures, fused forms, fallbacks, fuzz), where the launch spends its time
# (per-workgroup wall-clock stamps), the replay fused / in three launches with the oracle check, a kernel trace
export TMPDIR=/tmp
O=gpurun_out/r06_fused_sample; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fused_filter.py tests/test_k2_reference.py tests/test_gpu_sync_fallbacks.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py tests/test_gpu_scenes.py tests/test_gpu_occgrid.py -x -q 2>&1 | tail -6 > $O/tests.log
timeout 300 python tools/fuse_trace.py profiles/r02_sampling_state.npz 20 > $O/fuse_trace.log 2>&1
for f in 1 0; do
  NFA_FUSED_SAMPLE=$f python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 --check > $O/replay_f$f.log 2>&1
done
D=$(mktemp -d /tmp/ktXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 200 > /dev/null 2>&1
python tools/kernel_summary.py $D > $O/kstats_f1.txt 2>&1
for n in 6564 160000 1000000; do echo "rays=$n $(NFA_FUSED_SAMPLE=0 python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n 2>/dev/null | grep -o 'count [0-9.]* us  emit [0-9.]* us')"; done > $O/emit_sizes.txt
cat $O/tests.log $O/fuse_trace.log $O/replay_f1.log $O/replay_f0.log $O/emit_sizes.txt; grep nfa $O/kstats_f1.txt | head -3 | cut -c1-60,110-150
