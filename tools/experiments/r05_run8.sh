#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05h; export TMPDIR=/tmp
for v in norec base norec base; do
  lib=tools/_prof/libnerfacc_hip_$v.so; [ $v = base ] && lib=nerfacc_amd/libnerfacc_hip.so
  NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=$lib timeout 120 python tools/traverse_replay.py profiles/r02_sampling_state.npz 40 2>&1 | grep "^rays" | cut -c1-110 | sed "s/^/$v /"
done
