#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 100 python tools/traverse_replay.py profiles/r02_sampling_state.npz 10 --check 2>&1 | grep "oracle\|^rays" | cut -c1-120
for n in 6564 32000 160000 1000000; do
  for v in base emitbisect base emitbisect; do
    lib=tools/_prof/libnerfacc_hip_$v.so; [ $v = base ] && lib=nerfacc_amd/libnerfacc_hip.so
    NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=$lib timeout 120 python tools/traverse_replay.py profiles/r02_sampling_state.npz 12 --rays=$n 2>&1 | grep "^rays" | cut -c1-150 | sed "s/^/$v /"
  done
done
timeout 400 python -m pytest tests/test_k2_reference.py tests/test_gpu_grid.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
timeout 200 python tools/experiments/r04_emit_rb_scenes.py 256 noise 2>&1 | grep -v amdgpu | cut -c1-200
