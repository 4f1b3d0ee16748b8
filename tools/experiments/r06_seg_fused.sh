#!/bin/bash
# round 6, VERDICT r5 item 5a: the segment count kernel (several levels, cone_angle = 0) with the single launch's look-back + emit tail —
# fixtures and fuzz, then tools/multilevel_bench.py fused / in three launches
export TMPDIR=/tmp
O=gpurun_out/r06_seg_fused; mkdir -p $O
timeout 1200 python -m pytest tests/test_k2_reference.py tests/test_gpu_fuzz.py tests/test_gpu_testmode.py tests/test_gpu_sync_fallbacks.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6 > $O/tests.log
timeout 600 python tools/fuzz_levels.py 40 2609 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/tests.log
for f in 1 0 1 0; do echo "== NFA_FUSED_SAMPLE=$f"; NFA_FUSED_SAMPLE=$f timeout 200 python tools/multilevel_bench.py 4096 2>&1 | grep -v amdgpu.ids; done > $O/multilevel.txt
for f in 1 0; do echo "== NFA_FUSED_SAMPLE=$f 1024 rays"; NFA_FUSED_SAMPLE=$f timeout 200 python tools/multilevel_bench.py 1024 2>&1 | grep -v amdgpu.ids; done >> $O/multilevel.txt
cat $O/tests.log $O/multilevel.txt
