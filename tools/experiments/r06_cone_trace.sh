#!/bin/bash
# kernel trace of the several-level calls of tools/multilevel_bench.py (cone_angle 0.004 and the segment form): which kernel is the time
export TMPDIR=/tmp
for which in ML_ONLY_CONE ML_ONLY_LATTICE; do
  D=$(mktemp -d /tmp/ktXXXX)
  env $which=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ml -- python tools/multilevel_bench.py 4096 2> $D/err.txt | grep -v amdgpu
  python tools/kernel_summary.py $D | grep "nfa::" | sed 's/(.*)`/`/' | cut -c1-160 | head -8
  rm -rf $D
done
