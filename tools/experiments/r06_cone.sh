#!/bin/bash
# the cone-angle count pass after a change: its tests (fixtures, oracle, fuzz with a cone) and the several-level timings
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "cone or levels or multilevel or k2_reference" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
timeout 300 python tools/fuzz_levels.py --cone 40 2611 2>&1 | tail -1 | cut -c1-200
for n in 1024 2048 4096 8192; do ML_ONLY_CONE=1 ML_NO_CHECK=1 timeout 200 python tools/multilevel_bench.py $n 2>&1 | grep -v amdgpu | cut -c1-100; done
D=$(mktemp -d /tmp/ktXXXX)
ML_ONLY_CONE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ml -- python tools/multilevel_bench.py 4096 > /dev/null 2> $D/err.txt
python tools/kernel_summary.py $D | grep "nfa::" | sed 's/(.*)`/`/' | cut -c1-160 | head -3
rm -rf $D
