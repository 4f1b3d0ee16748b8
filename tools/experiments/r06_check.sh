#!/bin/bash
# quick GPU check of the modules touched since the last full suite
export TMPDIR=/tmp
O=gpurun_out/r06_check; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sync_fallbacks.py tests/test_gpu_fused_sampling.py tests/test_gpu_fused_filter.py tests/test_gpu_occgrid.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/tests.log
cat $O/tests.log
