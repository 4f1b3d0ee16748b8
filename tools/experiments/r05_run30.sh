#!/bin/bash
# (record of profiles/r05_streaming.md section 3; the variant libraries are render.hip built with -DNFA_VIS_EXP=<k> and linked with the other
#  objects into tools/_prof/libvis_exp<k>.so: 2 = no look-back, 4 = nothing staged / copied out; 1, 3, 7 existed for the first, per-wave-state form)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for k in 2 3 4 7; do echo "== exp $k"; NERFACC_AMD_LIB=$PWD/tools/_prof/libvis_exp$k.so timeout 60 python tools/experiments/r05_vis_onepass.py 24 2>&1 | grep "N=" | cut -c1-400; done
echo "== ctypes baseline"; NERFACC_AMD_BACKEND=ctypes timeout 60 python tools/experiments/r05_vis_onepass.py 24 2>&1 | grep "N=" | cut -c1-400
