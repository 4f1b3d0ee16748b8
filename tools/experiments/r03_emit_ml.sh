cd /root/repo; export TMPDIR=/tmp
for n in 1024 4096 16384; do
  for f in samples rays; do
    D=/tmp/ec_${n}_$f; mkdir -p $D
    echo "== $n $f"; NFA_EMIT=$f ML_ONLY_CONE=1 ML_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/multilevel_bench.py $n 2>&1 | grep levels | cut -c1-110
    python tools/kernel_summary.py $D | grep -E "traverse_" | cut -c1-150
  done
done
