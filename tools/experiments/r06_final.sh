#!/bin/bash
# round 6, the numbers of record in one GPU-box call: suite, smoke, bench line + kernel summary, counter passes, streaming sweeps,
# frame / micro / multi-level benchmarks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_suite.log 2>&1; echo "rc $?" >> gpurun_out/r06_suite.log
tail -4 gpurun_out/r06_suite.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_round.sh r06 bench > /dev/null 2>&1
bash tools/collect_round.sh r06 pmc > /dev/null 2>&1
bash tools/collect_round.sh r06 streaming > /dev/null 2>&1
timeout 200 python tools/roofline_sweep.py 18 gpurun_out/r06_stream18.md > gpurun_out/r06_stream18.txt 2>&1
timeout 300 bash tools/pmc_streaming_units.sh gpurun_out/r06_units 24 > gpurun_out/r06_units.txt 2>&1; rm -rf gpurun_out/r06_units/units
timeout 300 python tools/frame_bench.py 3 2>&1 | grep -v amdgpu > gpurun_out/r06_frame_bench.md
timeout 300 python tools/microbench.py gpurun_out/r06_microbench.md 2>&1 | grep -v amdgpu > gpurun_out/r06_microbench.txt
timeout 200 python tools/multilevel_bench.py 4096 2>&1 | grep -v amdgpu > gpurun_out/r06_multilevel.txt
bash tools/collect_round.sh r06 scenes > /dev/null 2>&1
head -c 700 gpurun_out/r06_bench_line.json; echo
grep "nfa::" gpurun_out/r06_bench_kernels_table.md | sed 's/(.*)`/`/' | cut -c1-150 | head -16
cat gpurun_out/r06_stream24.md | tail -14
tail -12 gpurun_out/r06_units.txt
cat gpurun_out/r06_frame_bench.md gpurun_out/r06_multilevel.txt
