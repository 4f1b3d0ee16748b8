#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in 3500 5000 6564 7500 8192; do
  for thr in auto 512 448 384 320 256; do
    if [ $thr = auto ]; then env="X=1"; else env="NFA_SPLIT_THR=$thr"; fi
    env $env timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 40 --rays=$n 2>&1 | grep "^rays" | cut -c1-100 | sed "s/^/thr=$thr /"
  done
done
timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check 2>&1 | grep oracle
timeout 300 python -m pytest tests/test_k2_reference.py -x -q -m gpu -p no:cacheprovider -k "lego_4k or lego_12k or m1_sphere" 2>&1 | tail -2
