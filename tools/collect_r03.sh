#!/bin/bash
# round-3 subset of tools/collect_profiles.sh (run on the GPU box from the repo root): the bench line, its kernel summary, and the
# counter passes over the sampling traversal of the frozen steady state (profiles/r02_sampling_state.npz) -> gpurun_out/r03_*
OUT=gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
python bench.py > $OUT/r03_bench_line.json 2> $OUT/r03_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-other-mode --no-aux --windows 1 > $OUT/r03_kt_line.json 2> $OUT/r03_kt.err
python tools/kernel_summary.py $OUT/r03_kt > $OUT/r03_bench_kernels_table.md
rm -rf $OUT/r03_kt
tools/pmc_traverse.sh profiles/r02_sampling_state.npz $OUT/r03_pmc 20 > /dev/null 2>&1
rm -rf $OUT/r03_pmc/*/*kernel_trace.csv $OUT/r03_pmc/*/*.db
ls $OUT $OUT/r03_pmc
