#!/bin/bash
# round 6: the reference eval loop (8192-ray chunks) after the single-launch forms — frame bench + the per-chunk profile
export TMPDIR=/tmp
O=gpurun_out/r06_eval; mkdir -p $O
timeout 600 python tools/frame_bench.py 3 > $O/frame_bench.md 2> $O/frame_bench.err
timeout 300 python tools/eval_chunk_profile.py > $O/eval_chunk_profile.txt 2>&1
cat $O/frame_bench.md; tail -40 $O/eval_chunk_profile.txt
