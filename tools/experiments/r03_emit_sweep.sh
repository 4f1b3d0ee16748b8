cd /root/repo
R="python tools/traverse_replay.py profiles/r02_sampling_state.npz 20"
for n in 1024 4096 6500 13000 32000 65536 160000 300000 1000000; do
  echo "== $n rays";    $R --rays=$n 2>&1 | tail -1
  echo "== $n samples"; NFA_EMIT=samples $R --rays=$n 2>&1 | tail -1
done
python tools/traverse_replay.py profiles/r02_sampling_state.npz 5 --check | tail -1
python tools/fuzz_campaign.py 30 902 2>&1 | tail -6
python tools/fuzz_levels.py 2>&1 | tail -4
python tools/fuzz_levels.py --cone 2>&1 | tail -4
