#!/bin/bash
# round 6, VERDICT r5 item 1d: the four streaming kernels at the bench's N, back to back on the same inputs / on inputs a kernel has
# just produced / with the caches flushed in between — kernel trace + counters (separate passes)
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_small_n; mkdir -p $O
for r in same produced evicted; do
  mkdir -p $O/$r
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/$r/trace -o t -- python $ROOT/tools/small_n_replay.py $r 50 > $O/$r/trace.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$r/pmc_tcc -o t -- python $ROOT/tools/small_n_replay.py $r 20 > $O/$r/pmc_tcc.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$r/pmc_fetch -o t -- python $ROOT/tools/small_n_replay.py $r 20 > $O/$r/pmc_fetch.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/$r/pmc_write -o t -- python $ROOT/tools/small_n_replay.py $r 20 > $O/$r/pmc_write.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/$r/pmc_sq -o t -- python $ROOT/tools/small_n_replay.py $r 20 > $O/$r/pmc_sq.log 2>&1)
  # keep the merge small: only the csv files we read
  find $O/$r -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*.log" -delete
done
cd $ROOT; head -3 $O/same/trace.log; python tools/small_n_table.py gpurun_out/r06_small_n | tee $O/table.md
find $O -name "*.csv" -size +3M -delete
