#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for k in 2 3 4 7; do echo "== exp $k"; NERFACC_AMD_LIB=$PWD/tools/_prof/libvis_exp$k.so timeout 60 python tools/experiments/r05_vis_onepass.py 24 2>&1 | grep "N=" | cut -c1-400; done
echo "== ctypes baseline"; NERFACC_AMD_BACKEND=ctypes timeout 60 python tools/experiments/r05_vis_onepass.py 24 2>&1 | grep "N=" | cut -c1-400
