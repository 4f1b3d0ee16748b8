#!/bin/bash
# (record of profiles/r05_streaming.md section 1: tools/_prof/libold_render.so = render.hip of commit 996dff3 linked with the current objects)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export NERFACC_AMD_LIB=$PWD/tools/_prof/libold_render.so; else unset NERFACC_AMD_LIB; export NERFACC_AMD_BACKEND=ctypes; fi
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05v -o kt -- python $GRAFT_REPO_ROOT/tools/roofline_sweep.py 24 > /dev/null 2>&1)
  echo "== $v"; python tools/kernel_summary.py gpurun_out/r05v | grep -i "visib\|rendering\|weight_" | cut -c1-60,100-160; rm -rf gpurun_out/r05v
done
