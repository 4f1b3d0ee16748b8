#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
NFA_FUZZ_SECONDS=200 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -k randomised -s 2>&1 | grep -v amdgpu | tail -6 | cut -c1-300
timeout 300 python tools/fuzz_campaign.py 400 2505 2>&1 | grep -v amdgpu | tail -3 | cut -c1-300
timeout 200 python tools/fuzz_levels.py 100 2506 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300
timeout 150 python tools/fuzz_levels.py 60 2507 --cone 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300
