#!/bin/bash
# (record of profiles/r05_streaming.md section 2: rendering_bwd with __launch_bounds__(256, 5 | 6) — tools/_prof/librbwd_mw<k>.so — and the E sweep)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
echo "== default (ctypes)"; NERFACC_AMD_BACKEND=ctypes timeout 100 python tools/roofline_sweep.py 24 2>&1 | grep "rendering_bwd\|rendering_fwd\|weight_bwd"
for k in 5 6; do echo "== minwaves $k"; NERFACC_AMD_LIB=$PWD/tools/_prof/librbwd_mw$k.so timeout 100 python tools/roofline_sweep.py 24 2>&1 | grep "rendering_bwd"; done
echo "== e=1"; NERFACC_AMD_BACKEND=ctypes NFA_E=1 timeout 100 python tools/roofline_sweep.py 24 2>&1 | grep "rendering_bwd\|rendering_fwd\|weight_bwd\|visib\|accumulate\|scan_keyed"
echo "== e=4"; NERFACC_AMD_BACKEND=ctypes NFA_E=4 timeout 100 python tools/roofline_sweep.py 24 2>&1 | grep "rendering_bwd\|rendering_fwd\|weight_bwd\|visib\|accumulate\|scan_keyed"
