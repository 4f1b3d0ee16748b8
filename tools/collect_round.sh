#!/bin/bash
# The judged numbers of a round, in one GPU-box call (from the repo root): tools/collect_round.sh r04  ->  gpurun_out/r04_*
#   * the bench line (defaults) and the rocprofv3 kernel summary of the same loop,
#   * counter passes (separate --pmc runs, never combined with other trace domains) over the sampling traversal of the frozen
#     steady state (profiles/r02_sampling_state.npz) -> <tag>_pmc/pmc_traverse.json, which bench.py reads for roofline.traffic,
#   * the scene sweep (tools/scene_sweep.py), the streaming kernels at 2^24 (tools/roofline_sweep.py), the emit pass against
#     the ray count (tools/experiments/r04_emit.sh).
# Every step under its own `timeout`; copy what should be judged into profiles/ afterwards (profiles/README.md says which file
# comes from which step).  Replaces tools/collect_profiles.sh (round 2) and tools/collect_r03.sh.
TAG=${1:-r04}; OUT=gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
only=${2:-all}
want() { [ "$only" = all ] || [ "$only" = "$1" ]; }
if want bench; then
  timeout 400 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-other-mode --no-aux --windows 1 > $OUT/${TAG}_kt_line.json 2> $OUT/${TAG}_kt.err
  python tools/kernel_summary.py $OUT/${TAG}_kt > $OUT/${TAG}_bench_kernels_table.md
  rm -rf $OUT/${TAG}_kt
fi
if want pmc; then
  timeout 400 tools/pmc_traverse.sh profiles/r02_sampling_state.npz $OUT/${TAG}_pmc_1m 6 --rays=1000000 > /dev/null 2>&1
  rm -rf $OUT/${TAG}_pmc_1m/*/*kernel_trace.csv $OUT/${TAG}_pmc_1m/*/*.db
  timeout 400 tools/pmc_traverse.sh profiles/r02_sampling_state.npz $OUT/${TAG}_pmc 20 > /dev/null 2>&1
  rm -rf $OUT/${TAG}_pmc/*/*kernel_trace.csv $OUT/${TAG}_pmc/*/*.db
fi
if want scenes; then timeout 300 python tools/scene_sweep.py $OUT/${TAG}_scene_sweep.md 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_scene_sweep.txt; fi
if want streaming; then timeout 300 python tools/roofline_sweep.py 24 $OUT/${TAG}_stream24.md 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_stream24.txt; fi
if want emit; then timeout 300 bash tools/experiments/r04_emit.sh 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_emit.txt; fi
ls $OUT | grep "^${TAG}_"
