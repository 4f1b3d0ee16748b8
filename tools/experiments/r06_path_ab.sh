#!/bin/bash
# round 6: path-only step with the single-launch forms on / off, wall time and a kernel trace of each
export TMPDIR=/tmp
O=gpurun_out/r06_path_ab; mkdir -p $O
timeout 600 python tools/path_ab.py 400 > $O/ab.log 2>&1
for f in 000 111; do
  D=$(mktemp -d /tmp/ktXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/path_ab.py 200 --forms=$f > /dev/null 2>&1
  python tools/kernel_summary.py $D > $O/kstats_$f.txt 2>&1
done
cat $O/ab.log; grep "nfa::" $O/kstats_000.txt $O/kstats_111.txt | cut -c1-200
