#!/bin/bash
# rocprofv3 passes over the sampling traversal of the bench's steady state (run on the GPU box from the repo root):
#   tools/pmc_traverse.sh profiles/r02_sampling_state.npz gpurun_out/pmc
# kernel trace + counters in SEPARATE passes (FETCH_SIZE and WRITE_SIZE cannot share one; each only with --kernel-trace).
set -e
STATE=$1; OUT=$2; REPS=${3:-20}; EXTRA=${4:-}      # EXTRA: further arguments of tools/traverse_replay.py, e.g. --rays=1000000
export TMPDIR=/tmp
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -o $name -- python tools/traverse_replay.py $STATE $REPS $EXTRA > $OUT/$name.log 2>&1 || tail -5 $OUT/$name.log; }
run trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run insts --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
run cycles --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU
python tools/pmc_json.py $STATE $OUT
