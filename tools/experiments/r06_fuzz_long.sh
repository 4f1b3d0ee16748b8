#!/bin/bash
# round 6, end-of-round differential fuzz at the last kernels (as every round since 3): the suite's randomised test for 200 s, the
# one-level campaign (incl. the single launch's window), several levels with and without cone angles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
NFA_FUZZ_SECONDS=200 timeout 600 python -m pytest tests/test_gpu_fuzz.py::test_randomised_fuzz_time_boxed -q -s 2>&1 | grep -i "fuzz seed\|passed\|failed\|MISMATCH\|seed" | tail -5 > gpurun_out/r06_fuzz_long.txt
timeout 1500 python tools/fuzz_campaign.py 400 2606 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r06_fuzz_long.txt
timeout 900 python tools/fuzz_levels.py 100 2607 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r06_fuzz_long.txt
timeout 900 python tools/fuzz_levels.py --cone 120 2608 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r06_fuzz_long.txt
cat gpurun_out/r06_fuzz_long.txt
