cd /root/repo; export TMPDIR=/tmp
for n in 6500 1000000; do for f in rays samples; do
  D=/tmp/em_${n}_$f; mkdir -p $D
  NFA_EMIT=$f rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n > /dev/null 2>&1
  echo "== $n $f"; python tools/kernel_summary.py $D | grep -E "traverse_|excl" | cut -c1-200
done; done
