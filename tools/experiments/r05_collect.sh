#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/collect_round.sh r05 > gpurun_out/r05_collect.txt 2>&1
timeout 300 python tools/frame_bench.py 3 2>&1 | grep -v amdgpu > gpurun_out/r05_frame_bench.md
timeout 300 python tools/microbench.py gpurun_out/r05_microbench.md 2>&1 | grep -v amdgpu > gpurun_out/r05_microbench.txt
timeout 200 python tools/multilevel_bench.py 4096 2>&1 | grep -v amdgpu > gpurun_out/r05_multilevel.txt
cat gpurun_out/r05_collect.txt | tail -20
cat gpurun_out/r05_frame_bench.md
cat gpurun_out/r05_multilevel.txt
head -c 1500 gpurun_out/r05_bench_line.json; echo
cat gpurun_out/r05_bench_kernels_table.md | head -40
