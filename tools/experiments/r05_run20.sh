#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in 6564 160000 1000000; do
  for v in base emitmb5 emitmb6 emitmb8 base; do
    lib=tools/_prof/libnerfacc_hip_$v.so; [ $v = base ] && lib=nerfacc_amd/libnerfacc_hip.so
    NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=$lib timeout 120 python tools/traverse_replay.py profiles/r02_sampling_state.npz 12 --rays=$n 2>&1 | grep "^rays" | cut -c1-105 | sed "s/^/$v /"
  done
done
