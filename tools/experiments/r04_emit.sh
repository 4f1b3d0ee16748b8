# round 4: emit pass at frame scale (closed-form lattice points per lane in the ray-group form).  HIP-event times of the count and
# emit C-ABI calls from tools/traverse_replay.py (bench steady state tiled to N rays); NFA_EMIT seeds the library's option table.
cd /root/repo
for n in 6500 13000 32000 160000 1000000; do for f in auto rays samples tiles; do
  echo "== $n emit=$f $(NFA_EMIT=$f python tools/traverse_replay.py profiles/r02_sampling_state.npz 20 --rays=$n | cut -d' ' -f1-20)"
done; done
