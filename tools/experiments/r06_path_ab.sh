#!/bin/bash
# round 6: path-only step with the single-launch forms on / off (digits: fused_sample fused_vis fold_fill; 101 = the defaults),
# wall time of every combination and a kernel trace of 000 / 101 / 111
export TMPDIR=/tmp
O=gpurun_out/r06_path_ab; mkdir -p $O
timeout 600 python tools/path_ab.py 400 > $O/ab.log 2>&1
for f in 000 101 111; do
  D=$(mktemp -d /tmp/ktXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/path_ab.py 200 --forms=$f --reps=1 > /dev/null 2>&1
  python tools/kernel_summary.py $D > $O/kstats_$f.txt 2>&1
done
cat $O/ab.log; grep "nfa::" $O/kstats_*.txt | cut -c1-200 | grep -v "bricks\|brick_dist"
