#!/bin/bash
# quick GPU check of the modules touched since the last full suite
export TMPDIR=/tmp
O=gpurun_out/r06_check; mkdir -p $O
NFA_TOL_AUDIT=1 timeout 1200 python -m pytest tests/test_gpu_occgrid.py tests/test_gpu_volrend.py tests/test_gpu_pdf.py tests/test_gpu_estimator.py tests/test_k2_reference.py tests/test_gpu_backends.py tests/test_host.py tests/test_gpu_fused_sampling.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/tests.log
cat $O/tests.log
