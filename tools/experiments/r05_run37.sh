#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 400 python bench.py > gpurun_out/r05_bench_stdout.txt 2> gpurun_out/r05_bench_stderr.txt; echo "rc $?"
echo "stdout lines: $(wc -l < gpurun_out/r05_bench_stdout.txt)"; head -c 200 gpurun_out/r05_bench_stdout.txt; echo
grep -c "RCCL version" gpurun_out/r05_bench_stderr.txt
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
