#!/bin/bash
# round 5, the numbers of record in one GPU-box call: suite, bench line + kernel summary, counter passes, smoke, streaming sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_suite.log 2>&1; echo "rc $?" >> gpurun_out/r05_suite.log
tail -4 gpurun_out/r05_suite.log | cut -c1-200
bash tools/collect_round.sh r05 bench > /dev/null 2>&1
bash tools/collect_round.sh r05 pmc > /dev/null 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_round.sh r05 streaming > /dev/null 2>&1
timeout 200 python tools/roofline_sweep.py 18 gpurun_out/r05_stream18.md > gpurun_out/r05_stream18.txt 2>&1
timeout 300 bash tools/pmc_streaming_units.sh gpurun_out/r05_units 24 > gpurun_out/r05_units.txt 2>&1; rm -rf gpurun_out/r05_units/units
head -c 600 gpurun_out/r05_bench_line.json; echo
grep "traverse_\|brick_dist\|visibility\|rendering_" gpurun_out/r05_bench_kernels_table.md | cut -c1-150 | head -14
cat gpurun_out/r05_stream24.md | tail -14
tail -12 gpurun_out/r05_units.txt
