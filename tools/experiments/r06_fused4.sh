#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_fused4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fused_filter.py tests/test_gpu_fill_fold.py tests/test_gpu_visibility_onepass.py -q 2>&1 | tail -25 > $O/tests.log
timeout 600 python tools/path_ab.py 400 > $O/ab.log 2>&1
for f in 111 121; do
  D=$(mktemp -d /tmp/ktXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/path_ab.py 200 --forms=$f > /dev/null 2>&1
  python tools/kernel_summary.py $D > $O/kstats_$f.txt 2>&1
done
cat $O/tests.log $O/ab.log; grep "nfa::" $O/kstats_1*.txt | cut -c1-200 | grep -v "bricks\|brick_dist"
