#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in 3500 5000 6564 8192; do
  for p in 32 16; do
    NFA_SPLIT_P=$p timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 40 --rays=$n 2>&1 | grep "^rays" | cut -c1-100 | sed "s/^/P=$p /"
  done
done
NFA_SPLIT_P=32 timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check 2>&1 | grep "oracle\|rays"
timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check 2>&1 | grep "oracle\|rays"
timeout 300 python -m pytest tests/test_k2_reference.py -x -q -m gpu -p no:cacheprovider -k "lego_4k or lego_12k or m1_sphere or m1_noise" 2>&1 | tail -2
