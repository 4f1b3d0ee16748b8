#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_volrend.py tests/test_gpu_scan.py tests/test_gpu_tiles.py tests/test_gpu_visibility_onepass.py tests/test_gpu_backends.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 200 python tools/roofline_sweep.py 24 2>&1 | grep "kernel\|visib" | grep -v "tuned\|torch copy"
timeout 200 python tools/roofline_sweep.py 18 2>&1 | grep "kernel\|visib" | grep -v "tuned\|torch copy"
