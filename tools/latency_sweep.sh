#!/bin/bash
# tools/latency_sweep.sh run|cold   (on the GPU box, from the repo root; env NFA_TILE / NFA_E pass through)
export TMPDIR=/tmp
D=$(mktemp -d /tmp/latXXXX)
rocprofv3 --kernel-trace --output-format csv -d $D -o lat -- python tools/latency_sweep.py $1 > $D/log.txt 2>&1
python tools/latency_sweep.py report $(find $D -name "*kernel_trace.csv" | head -1)
rm -rf $D
