#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2; do timeout 200 python tools/multilevel_bench.py 4096 2>&1 | grep -v amdgpu; done
NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=nerfacc_amd/libnerfacc_hip.so timeout 300 python tools/experiments/r05_count_ab.py new --quick 2>/dev/null | grep "4 x 128" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(f\"{d['workload']:24s} {d['rays']:8d} {d.get('form','auto'):12s} count {d['count_us']:8.1f} emit {d['emit_us']:7.1f}\")"
timeout 400 python -m pytest tests/test_k2_reference.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider -k "levels or near_far or degenerate or non_cubic or cone or inplane or fuzz" 2>&1 | tail -2
