#!/bin/bash
# round 6: rendering_bwd without the ext-gradient payload (75 VGPRs instead of 98): tests, 2^24 sweep, small-N kernel times
export TMPDIR=/tmp
O=gpurun_out/r06_bwd; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_volrend.py tests/test_gpu_semantics.py tests/test_gpu_training.py tests/test_gpu_tiles.py tests/test_gpu_backends.py -x -q 2>&1 | tail -4 > $O/tests.log
timeout 300 python tools/roofline_sweep.py 24 $O/stream24.md 2>&1 | grep -v amdgpu.ids > $O/stream24.txt
D=$(mktemp -d /tmp/ktXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/path_ab.py 200 --forms=101 --reps=1 > /dev/null 2>&1
python tools/kernel_summary.py $D > $O/kstats_101.txt 2>&1
cat $O/tests.log; grep "kernel\|visib\|tuned" $O/stream24.txt | cut -c1-150; grep "nfa::" $O/kstats_101.txt | sed 's/(.*)`/`/' | cut -c1-120 | head -6
