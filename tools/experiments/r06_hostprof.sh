#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_hostprof; mkdir -p $O
timeout 300 python tools/path_host_profile.py --cprofile > $O/prof.txt 2>&1
grep -v amdgpu.ids $O/prof.txt | head -70
