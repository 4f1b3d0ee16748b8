cd /root/repo; export TMPDIR=/tmp
for n in 4096 6500 9000 13000 20000 32000; do for f in rays samples; do
  D=/tmp/es_${n}_$f; mkdir -p $D
  NFA_EMIT=$f rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n > /dev/null 2>&1
  echo "== $n $f $(python tools/kernel_summary.py $D | grep -E 'emit' | cut -d'|' -f4)"
done; done
for n in 2048 4096 8192; do for f in rays samples; do
  D=/tmp/esm_${n}_$f; mkdir -p $D
  NFA_EMIT=$f ML_ONLY_LATTICE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/multilevel_bench.py $n > /dev/null 2>&1
  echo "== levels $n $f $(python tools/kernel_summary.py $D | grep -E 'emit' | cut -d'|' -f4)"
done; done
