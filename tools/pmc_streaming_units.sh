#!/bin/bash
# Which unit do the streaming kernels wait for?  One rocprofv3 counter pass (8 SQ slots, with --kernel-trace only) over tools/stream_replay.py:
#   tools/pmc_streaming_units.sh <out_dir> [log2_N]        (on the GPU box, from the repo root)
# SQ_WAVE_CYCLES = WAIT_ANY (wave parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY, in quad-cycles
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -e
OUT=$1; LOGN=${2:-24}
export TMPDIR=/tmp
mkdir -p $OUT
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d $ROOT/$OUT/units -o units -- python $ROOT/tools/stream_replay.py $LOGN 3 > $ROOT/$OUT/units.log 2>&1) || tail -5 $OUT/units.log
python tools/pmc_streaming_units_table.py $OUT $LOGN
