# round 4: rays per wave of the tile emit form (NFA_EMIT_RB = log2) against the ray count; emit time from the call's HIP events
cd /root/repo
for n in 1024 2048 4096 6500 13000 32000 160000 1000000; do
  line="rays $n:"
  for rb in 0 1 2 3 4 5 6; do
    e=$(NFA_EMIT_RB=$rb python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n 2>/dev/null | sed 's/.*emit \([0-9.]*\) us.*/\1/')
    line="$line rb=$rb $e"
  done
  echo "$line"
done
