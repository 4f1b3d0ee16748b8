#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05d; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; local t0=$(date +%s); timeout -k 5 $secs "$@" > gpurun_out/r05d/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/r05d/summary.txt; }
step ab_new 500 env NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=nerfacc_amd/libnerfacc_hip.so python tools/experiments/r05_count_ab.py new
grep "frame" gpurun_out/r05d/ab_new.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(f\"{d['workload']:24s} {d['rays']:8d} {d.get('form','auto'):16s} count {d['count_us']:8.1f} emit {d['emit_us']:7.1f} samples {d['samples']}\")"
step suite 1300 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_multirank.py -p no:cacheprovider
tail -5 gpurun_out/r05d/suite.log | cut -c1-300
step multirank 400 python -m pytest tests/test_gpu_bench_multirank.py -q -m gpu -x -p no:cacheprovider
tail -5 gpurun_out/r05d/multirank.log | cut -c1-300
