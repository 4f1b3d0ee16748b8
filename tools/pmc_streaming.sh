#!/bin/bash
# HBM-side traffic of the streaming kernels (run on the GPU box from the repo root):  tools/pmc_streaming.sh <out_dir> [log2_N]
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), each only with --kernel-trace.
set -e
OUT=$1; LOGN=${2:-24}
export TMPDIR=/tmp
mkdir -p $OUT
ROOT=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/$c -o $c -- python $ROOT/tools/stream_replay.py $LOGN 3 > $ROOT/$OUT/$c.log 2>&1) || tail -5 $OUT/$c.log
done
python tools/pmc_streaming_table.py $OUT $LOGN
