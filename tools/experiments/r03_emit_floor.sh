cd /root/repo; export TMPDIR=/tmp
for v in none emitdbg1 emitdbg2; do
  D=/tmp/ef_$v; mkdir -p $D
  if [ $v = none ]; then E=""; else E="NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=tools/_prof/libnerfacc_hip_$v.so"; fi
  echo "== $v"
  env $E NFA_EMIT=rays ML_ONLY_CONE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/multilevel_bench.py 4096 > /dev/null 2>&1
  python tools/kernel_summary.py $D | grep -E "emit" | cut -c1-150
  D=/tmp/ef2_$v; mkdir -p $D
  env $E NFA_EMIT=rays rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 > /dev/null 2>&1
  python tools/kernel_summary.py $D | grep -E "emit" | cut -c1-150
done
