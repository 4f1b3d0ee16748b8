#!/bin/bash
# round 6: fused sampling launch — its test module, then where the launch spends its time (per-workgroup wall-clock stamps)
export TMPDIR=/tmp
O=gpurun_out/r06_fused2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_sampling.py -x -q 2>&1 | tail -15 > $O/tests.log
timeout 300 python tools/fuse_trace.py profiles/r02_sampling_state.npz 20 > $O/fuse_trace.log 2>&1
cat $O/tests.log $O/fuse_trace.log
